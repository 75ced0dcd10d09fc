// probe: HIP IPC between two processes on ONE GPU (fork before any HIP call), uncached allocation, flag polling from a kernel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <sys/wait.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("rank %d: %s -> %s\n", rank, #x, hipGetErrorString(e_)); fflush(stdout); _exit(2); } } while (0)
__global__ void k_push(unsigned long long* peer, unsigned long long* flag_peer, int rank, unsigned long long q) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 1024) peer[rank * 1024 + i] = q * 1000 + rank * 100 + (i & 7);
}
__global__ void k_signal(unsigned long long* flag_peer, int rank, unsigned long long q) {
  __hip_atomic_store(&flag_peer[rank], q, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_wait_sum(unsigned long long* mine, unsigned long long* flags, int world, unsigned long long q, unsigned long long* out, int* status) {
  __shared__ int ok;
  if (threadIdx.x == 0) {
    ok = 1;
    for (int r = 0; r < world; ++r) {
      long long spins = 0;
      while (__hip_atomic_load(&flags[r], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < q) { if (++spins > 200000000ll) { ok = 0; break; } __builtin_amdgcn_s_sleep(16); }
    }
    *status = ok;
  }
  __syncthreads();
  if (!ok) return;
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) { unsigned long long s = 0; for (int r = 0; r < world; ++r) s += __builtin_nontemporal_load(&mine[r * 1024 + i]); out[i] = s; }
}
int main() {
  int pfd[2][2]; pipe(pfd[0]); pipe(pfd[1]);
  const int world = 2;
  pid_t pid = fork();
  const int rank = pid == 0 ? 1 : 0;
  CK(hipSetDevice(0));
  unsigned long long *buf, *flags, *out; int* status;
  CK(hipExtMallocWithFlags((void**)&buf, world * 1024 * 8, hipDeviceMallocUncached));
  CK(hipExtMallocWithFlags((void**)&flags, 64, hipDeviceMallocUncached));
  CK(hipMalloc(&out, 1024 * 8)); CK(hipMalloc(&status, 4));
  CK(hipMemset(buf, 0, world * 1024 * 8)); CK(hipMemset(flags, 0, 64));
  hipIpcMemHandle_t h[2], peer[2];
  CK(hipIpcGetMemHandle(&h[0], buf)); CK(hipIpcGetMemHandle(&h[1], flags));
  // exchange handles through the pipes
  write(pfd[rank][1], h, sizeof(h)); read(pfd[1 - rank][0], peer, sizeof(peer));
  unsigned long long *pbuf, *pflags;
  CK(hipIpcOpenMemHandle((void**)&pbuf, peer[0], hipIpcMemLazyEnablePeerAccess));
  CK(hipIpcOpenMemHandle((void**)&pflags, peer[1], hipIpcMemLazyEnablePeerAccess));
  int bad = 0;
  for (unsigned long long q = 1; q <= 50; ++q) {
    // push into own and peer buffer, signal both, wait, sum
    hipLaunchKernelGGL(k_push, dim3(4), dim3(256), 0, 0, buf, flags, rank, q);
    hipLaunchKernelGGL(k_push, dim3(4), dim3(256), 0, 0, pbuf, pflags, rank, q);
    hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, 0, flags, rank, q);
    hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, 0, pflags, rank, q);
    hipLaunchKernelGGL(k_wait_sum, dim3(1), dim3(256), 0, 0, buf, flags, world, q, out, status);
    unsigned long long o[1024]; int st;
    CK(hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost)); CK(hipMemcpy(&st, status, 4, hipMemcpyDeviceToHost));
    if (!st) { printf("rank %d: exchange %llu timed out\n", rank, q); bad = 1; break; }
    for (int i = 0; i < 1024; ++i) if (o[i] != 2 * q * 1000 + 100 + 2 * (i & 7)) { printf("rank %d: q %llu i %d got %llu\n", rank, q, i, o[i]); bad = 1; break; }
    if (bad) break;
  }
  printf("rank %d: %s\n", rank, bad ? "FAILED" : "50 exchanges ok"); fflush(stdout);
  CK(hipIpcCloseMemHandle(pbuf)); CK(hipIpcCloseMemHandle(pflags));
  if (rank == 0) { int stt; waitpid(pid, &stt, 0); }
  return bad;
}
