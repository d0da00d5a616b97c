// Leaves recognisable garbage behind on every CU: NaN in (almost) every vector register of the wave and in all 160 KB of LDS.
// A kernel that reads a register or an LDS word it never wrote picks these up instead of whatever the previous wave happened to
// leave (round 4: tools/pfn_race_probe3.sh -- is the pillar feature net backward's cross-process failure an uninitialised read?).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/poison tools/poison.hip          stand-alone neighbour process: poison <seconds> [what]
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/bin/libpoison.so tools/poison.hip     poison_launch(stream, what) for a side stream
// what: 1 = LDS, 2 = VGPRs, 4 = SGPR-visible state (M0), 7 = all
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#define R8(a) "v_mov_b32 v" #a "0, %0\n v_mov_b32 v" #a "1, %0\n v_mov_b32 v" #a "2, %0\n v_mov_b32 v" #a "3, %0\n v_mov_b32 v" #a "4, %0\n v_mov_b32 v" #a "5, %0\n v_mov_b32 v" #a "6, %0\n v_mov_b32 v" #a "7, %0\n v_mov_b32 v" #a "8, %0\n v_mov_b32 v" #a "9, %0\n"

__global__ __launch_bounds__(256) void poison_kernel(int what, float* sink) {
  extern __shared__ float lds[];
  const float nanv = __builtin_nanf("");
  if (what & 1)
    for (int i = threadIdx.x; i < 40960; i += 256) lds[i] = nanv;
  if (what & 2) {
    // v10 .. v249: 240 registers written with NaN (the clobber list makes the compiler allocate them)
    asm volatile(R8(1) R8(2) R8(3) R8(4) R8(5) R8(6) R8(7) R8(8) R8(9) R8(10) R8(11) R8(12) R8(13) R8(14) R8(15) R8(16) R8(17) R8(18) R8(19) R8(20) R8(21) R8(22) R8(23) R8(24)
                 :: "v"(nanv)
                 : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29",
                   "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49",
                   "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69",
                   "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89",
                   "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109",
                   "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129",
                   "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149",
                   "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169",
                   "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189",
                   "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209",
                   "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229",
                   "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249");
  }
  __syncthreads();
  if (sink && lds[threadIdx.x] == 1.f) sink[0] = 1.f;    // (keeps the LDS stores alive)
}

static void launch(hipStream_t s, int what) {
  static bool once = false;
  if (!once) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(poison_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    once = true;
  }
  // 2048 workgroups x 160 KB of LDS: one per CU at a time, eight rounds over the chip
  hipLaunchKernelGGL(poison_kernel, dim3(2048), dim3(256), 163840, s, what, (float*)nullptr);
}

extern "C" int poison_launch(void* stream, int what) {
  launch(reinterpret_cast<hipStream_t>(stream), what);
  return (int)hipGetLastError();
}

#ifndef POISON_LIB
int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 60.0;
  const int what = argc > 2 ? atoi(argv[2]) : 7;
  const auto t0 = std::chrono::steady_clock::now();
  long n = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
    for (int i = 0; i < 50; ++i) launch(nullptr, what);
    hipDeviceSynchronize();
    n += 50;
  }
  printf("poison: %ld launches, last error %d\n", n, (int)hipGetLastError());
  return 0;
}
#endif
