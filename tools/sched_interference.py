"""How much does a co-running kernel that pins a few CUs (what an RCCL all-reduce does during the data-parallel
backward) delay the persistent GEMMs, with static vs dynamic tile scheduling?  Run once per NVIT_GEMM_SCHED value.
The hog is a torch kernel is not controllable enough, so it is a tiny HIP module compiled at run time with hipcc."""
import ctypes, os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvit_amd import ops

SRC = r'''
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(512) void hog(long long ticks, int* sink) {
  extern __shared__ char smem[];
  smem[threadIdx.x] = 1;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (threadIdx.x == 0 && ticks < 0) sink[0] = smem[1];
}
extern "C" int launch_hog(int wgs, long long ticks, int* sink, void* stream) {
  static bool set = false;
  if (!set) { hipFuncSetAttribute((const void*)hog, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024); set = true; }
  hipLaunchKernelGGL(hog, dim3(wgs), dim3(512), 100 * 1024, (hipStream_t)stream, ticks, sink);
  return (int)hipGetLastError();
}
extern "C" int clock_khz() { int r = 0; hipDeviceGetAttribute(&r, hipDeviceAttributeWallClockRate, 0); return r; }
'''
d = tempfile.mkdtemp()
open(os.path.join(d, "hog.hip"), "w").write(SRC)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(d, "hog.hip"),
                       "-o", os.path.join(d, "libhog.so")], stderr=subprocess.DEVNULL)
hog = ctypes.CDLL(os.path.join(d, "libhog.so"))
hog.launch_hog.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
khz = hog.clock_khz()
dev = "cuda:0"
M, N, K = 100352, 768, 768
A = torch.randn(M, K, device=dev).bfloat16()
B = torch.randn(N, K, device=dev).bfloat16()
out = torch.empty(M, N, device=dev, dtype=torch.float32)
sink = torch.zeros(4, device=dev, dtype=torch.int32)
side = torch.cuda.Stream()


def run(hog_wgs, hog_us, n=300, every=6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        if hog_wgs and i % every == 0:
            hog.launch_hog(hog_wgs, int(khz * hog_us / 1000), sink.data_ptr(), side.cuda_stream)
        ops.gemm_nt(A, B, M, N, K, out=out)
    torch.cuda.current_stream().synchronize()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt / n * 1e6


for _ in range(2):
    run(0, 0, 50)
base = run(0, 0)
print(f"sched={os.environ.get('NVIT_GEMM_SCHED', 'static')}: no interference {base:.1f} us/GEMM", flush=True)
for wgs, us in ((16, 300), (32, 300), (64, 300)):
    t = run(wgs, us)
    # the hog holds wgs CUs for `us` out of every 6 GEMMs: ideal cost = its share of the machine
    ideal = base * (1 + (wgs / 256) * us / (6 * base))
    print(f"   hog {wgs} CUs x {us} us every 6 GEMMs: {t:.1f} us/GEMM (+{(t / base - 1) * 100:.1f} %; share-of-machine ideal +{(ideal / base - 1) * 100:.1f} %)", flush=True)
