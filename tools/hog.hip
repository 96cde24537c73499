// Test utility (tools/hog_ab.py): a kernel that holds `blocks` workgroups of 256 threads for ~`cycles` shader clocks - a stand-in
// for a collective's long-running workgroups when measuring how compute kernels tolerate losing CUs.
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(256) void hog_k(long long cycles, int* sink) {
  const long long t0 = wall_clock64();
  long long t = t0;
  while (t - t0 < cycles) t = wall_clock64();
  if (t == 0x7fffffffffffffffLL) *sink = 1;
}
extern "C" int hog_launch(int blocks, long long cycles, int* sink, void* stream) {
  hipLaunchKernelGGL(hog_k, dim3(blocks), dim3(256), 0, (hipStream_t)stream, cycles, sink);
  return (int)hipGetLastError();
}
