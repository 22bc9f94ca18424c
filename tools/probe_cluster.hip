// Probe: cost and correctness of an intra-XCD cluster barrier + hand-off on MI355X.
// 256 workgroups (1 per CU, forced by a large LDS request), cluster = blockIdx % 8 (observed to be
// the XCD).  Each iteration: every WG writes a slab (plain stores), cluster barrier (device-scope
// atomic counter, relaxed sc1 polling), then reads ALL slabs of its cluster with sc1 loads and
// verifies them.  Reports XCC-id census, errors, and time per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef unsigned int u32;
typedef __attribute__((address_space(1))) u32 gu32;

__device__ __forceinline__ u32 ld_sc1(const u32* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifndef BARRIER_SLEEP
#define BARRIER_SLEEP 1
#endif
__device__ __forceinline__ bool cluster_barrier(u32* counter, u32 target, u32* abort_flag) {
  __syncthreads();
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const u32 before = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (before + 1 < target && ld_sc1(counter) < target) {
      if (BARRIER_SLEEP) __builtin_amdgcn_s_sleep(BARRIER_SLEEP);
      if (++spins > (1u << 22)) { __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      if ((spins & 1023u) == 0 && ld_sc1(abort_flag)) break;
    }
  }
  __syncthreads();
  return true;
}

__global__ __launch_bounds__(512) void k_probe(u32* slabs, u32* counters, u32* abort_flag, u32* xcc, u32* errors,
                                               unsigned long long* cycles, int iters, int slab_words) {
  extern __shared__ unsigned char lds[];
  if (threadIdx.x == 0) lds[0] = 1;
  const int cluster = blockIdx.x & 7, rank = blockIdx.x >> 3;   // 32 ranks per cluster
  if (threadIdx.x == 0) {
    u32 id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc[blockIdx.x] = id & 0xf;
  }
  u32* my = slabs + ((size_t)cluster * 32 + rank) * slab_words;
  u32* ctr = counters + cluster * 64;
  u32 err = 0;
  unsigned long long t0 = 0;
  for (int it = 0; it < iters; ++it) {
    if (it == 8 && threadIdx.x == 0) t0 = wall_clock64();
    for (int i = threadIdx.x; i < slab_words; i += blockDim.x) my[i] = (u32)(it * 1000003 + rank * 131 + i);
    cluster_barrier(ctr, (u32)(32 * (2 * it + 1)), abort_flag);
#if defined(READ_X4)
    // 16-byte sc1 loads (L1 bypassed, L2 served): what an MFMA operand stream would use
    for (int r = 0; r < 32; ++r) {
      const u32* other = slabs + ((size_t)cluster * 32 + r) * slab_words;
      for (int i = threadIdx.x * 4; i < slab_words; i += blockDim.x * 4) {
        typedef u32 u32x4 __attribute__((ext_vector_type(4)));
        u32x4 v;
        const u32* pp = other + i;
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(pp) : "memory");
        for (int k = 0; k < 4; ++k) err += v[k] != (u32)(it * 1000003 + r * 131 + i + k);
      }
    }
#elif !defined(SKIP_READ)
    for (int r = 0; r < 32; ++r) {
      const u32* other = slabs + ((size_t)cluster * 32 + r) * slab_words;
      for (int i = threadIdx.x; i < slab_words; i += blockDim.x)
        err += ld_sc1(other + i) != (u32)(it * 1000003 + r * 131 + i);
    }
#endif
    cluster_barrier(ctr, (u32)(32 * (2 * it + 2)), abort_flag);   // nobody overwrites before all have read
    if (ld_sc1(abort_flag)) break;
  }
  if (threadIdx.x == 0) cycles[blockIdx.x] = wall_clock64() - t0;
  atomicAdd(errors, err);
}

int main() {
  const int iters = 2000;
  for (int slab_words : {64, 512, 4096}) {
    u32 *slabs, *counters, *abort_flag, *xcc, *errors; unsigned long long* cycles;
    CK(hipMalloc(&slabs, (size_t)256 * slab_words * 4)); CK(hipMalloc(&counters, 8 * 64 * 4));
    CK(hipMalloc(&abort_flag, 4)); CK(hipMalloc(&xcc, 256 * 4)); CK(hipMalloc(&errors, 4)); CK(hipMalloc(&cycles, 256 * 8));
    CK(hipMemset(counters, 0, 8 * 64 * 4)); CK(hipMemset(abort_flag, 0, 4)); CK(hipMemset(errors, 0, 4));
    CK(hipMemset(slabs, 0, (size_t)256 * slab_words * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    hipLaunchKernelGGL(k_probe, dim3(256), dim3(512), 100 * 1024, 0, slabs, counters, abort_flag, xcc, errors, cycles, iters, slab_words);
    CK(hipDeviceSynchronize());
    std::vector<u32> hx(256); u32 herr, habort; std::vector<unsigned long long> hc(256);
    CK(hipMemcpy(hx.data(), xcc, 1024, hipMemcpyDeviceToHost)); CK(hipMemcpy(&herr, errors, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&habort, abort_flag, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hc.data(), cycles, 2048, hipMemcpyDeviceToHost));
    int mismatch = 0;
    for (int b = 0; b < 256; ++b) mismatch += hx[b] != hx[b & 7];
    unsigned long long mx = 0; for (auto c : hc) mx = c > mx ? c : mx;
    printf("slab %5d B: xcc ids of blocks 0..7 = %u %u %u %u %u %u %u %u; blocks off their cluster's XCD: %d; errors %u; abort %u; "
           "%.2f us per iteration (2 barriers + write + read-all)\n", slab_words * 4, hx[0], hx[1], hx[2], hx[3], hx[4], hx[5], hx[6], hx[7],
           mismatch, herr, habort, (double)mx / 100.0 / (iters - 8));
  }
  return 0;
}
