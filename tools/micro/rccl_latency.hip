// Latency of ncclAllReduce(int64, sum) for the window sizes of the table exchange, ONE rank (what a 1-GPU box can measure: the
// fixed cost of an RCCL collective — launch path, work FIFO, kernel prologue — without any link time).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/rccl_latency.hip -lrccl -o tools/micro/rccl_latency && tools/micro/rccl_latency
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
__global__ void k_nop(long long* p) { if (threadIdx.x == 9999) p[0] = 1; }
int main() {
  ncclUniqueId id; ncclGetUniqueId(&id);
  ncclComm_t c; fflush(stdout); int so = dup(1); dup2(2, 1); ncclCommInitRank(&c, 1, id, 0); fflush(stdout); dup2(so, 1);
  hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  long long* d; (void)hipMalloc(&d, 11340 * 8); (void)hipMemset(d, 0, 11340 * 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (size_t n : {(size_t)1, (size_t)2268, (size_t)11340}) {
    for (int i = 0; i < 5; ++i) ncclAllReduce(d, d, n, ncclInt64, ncclSum, c, st);
    (void)hipStreamSynchronize(st);
    // (a) back to back on the stream, between two small kernels (as the exchange sits between the flush and the fold)
    const int R = 50;
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < R; ++i) { hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, st, d); ncclAllReduce(d, d, n, ncclInt64, ncclSum, c, st); hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, st, d); }
    (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < R; ++i) { hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, st, d); hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, st, d); }
    (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
    float ms0; (void)hipEventElapsedTime(&ms0, e0, e1);
    // (b) host round trip: enqueue one collective on an idle stream and wait for it
    double host = 0;
    for (int i = 0; i < R; ++i) {
      (void)hipStreamSynchronize(st);
      const auto t0 = std::chrono::steady_clock::now();
      ncclAllReduce(d, d, n, ncclInt64, ncclSum, c, st);
      while (hipStreamQuery(st) == hipErrorNotReady) {}
      host += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
    printf("{\"ranks\": 1, \"int64_words\": %zu, \"bytes\": %zu, \"device_us_per_allreduce_between_two_kernels\": %.1f, \"two_kernels_alone_us\": %.1f, \"host_round_trip_us\": %.1f}\n",
           n, n * 8, (ms - ms0) * 1e3 / R, ms0 * 1e3 / R, host / R);
  }
  ncclCommDestroy(c);
  return 0;
}
