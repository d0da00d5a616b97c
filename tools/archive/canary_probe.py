"""LDS / register canaries (tools/canary.hip) on one stream while a neighbour THREAD of this process runs library kernels on another:
does a kernel of this library write outside its own LDS allocation or registers?   (round 4, after tools/pfn_race_probe7.sh named
df_gru_decoder_fwd as the neighbour the SLP-built pillar kernels react to)
    DF_NB_ONLY=df_gru python tools/canary_probe.py infer [launches] [lds_bytes]"""
import ctypes, os, sys, threading, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import pfn_neighbour
kind = sys.argv[1] if len(sys.argv) > 1 else "infer"
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
lds_bytes = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
dev = torch.device("cuda", 0)
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libcanary.so"))
bad = torch.zeros(2, dtype=torch.int32, device=dev)
detail = torch.zeros(4, dtype=torch.int32, device=dev)
stop, count = threading.Event(), [0]


def nb():
    torch.cuda.set_device(dev)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        step = pfn_neighbour.make(kind, dev)
        while not stop.is_set():
            step()
            count[0] += 1
            if count[0] % 8 == 0:
                st.synchronize()


th = None
if kind != "none":
    th = threading.Thread(target=nb, daemon=True)
    th.start()
    time.sleep(8.0)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for i in range(launches):
        lib.canary_launch(ctypes.c_void_p(s.cuda_stream), ctypes.c_void_p(bad.data_ptr()), lds_bytes, 40, 64, ctypes.c_void_p(detail.data_ptr()))
        if i % 64 == 63:
            s.synchronize()
s.synchronize()
stop.set()
if th is not None:
    th.join(10.0)
b = bad.tolist()
print(f"canary, neighbour thread {kind} (only /{os.environ.get('DF_NB_ONLY', '.')}/, {count[0]} steps), {launches} launches x 64 workgroups x {lds_bytes} B LDS: "
      f"{b[0]} LDS words changed, {b[1]} register values changed; last LDS hit (word, got, want, block) = {[hex(x & 0xffffffff) for x in detail.tolist()]}")
