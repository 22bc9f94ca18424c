// handoff_bench.hip -- what one all-to-all hand-off between the 32 workgroups of an XCD costs,
// two ways.  Standalone (hipcc --offload-arch=gfx950 handoff_bench.hip -o handoff_bench).
//
// The one-launch decode (k_decode_resident) hands a [rows x 512] activation from 32 producer
// workgroups (16 features each) to the same 32 workgroups as consumers, three times per step.
// Variant A is what the kernel does today: stores, s_waitcnt vmcnt(0), an L2 atomic arrival, sc1
// polling of the counter, then the operand loads.  Variant B signals in-band: the buffer holds a
// signalling-NaN pattern no arithmetic can produce, consumers poll the DATA, producers reset their
// region two stages later (three buffers in rotation).  No fence, no counter.
//
// Each stage: every wave loads its share of the rows (the MFMA B-operand pattern: 4 k-blocks per
// wave, 16 rows x 4 quads per load), folds them into a checksum, and the workgroup writes
// rows x 16 new values derived from it.  Both variants must end with the same checksum.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define WG 32            // workgroups per cluster (one XCD)
#define NCL 8            // clusters
#define KB 32            // k-blocks = producers
#define SENT 0x7f80deadu // signalling NaN: never the result of an arithmetic instruction

#define HIPCHK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ u32x4 load_sc1(const u32x4* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

__device__ __forceinline__ u32x4 issue_sc1(const u32x4* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// buffer layout per cluster: [3 buffers][tile][kb][16 rows][16 floats]  (tile-major like hst)
__device__ __forceinline__ size_t unit(int buf, int tiles, int tile, int kb) {
  return (((size_t)buf * tiles + tile) * KB + kb) * 256;
}

template <int VARIANT, int TILES, bool CANARY>
__global__ __launch_bounds__(512) void k_handoff(float* __restrict__ data, uint32_t* __restrict__ bar, int tiles, int stages,
                                                 int busy, float* __restrict__ out, uint32_t* __restrict__ spins_out) {
  const int cluster = blockIdx.x % NCL, rank = blockIdx.x / NCL;
  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  float* base = data + (size_t)cluster * 3 * tiles * KB * 256;
  uint32_t* ctr = bar + cluster * 32;
  __shared__ float s_part[8];
  __shared__ int s_flag;
  float carry = (float)rank;
  uint32_t spins = 0;
  for (int s = 0; s < stages; ++s) {
    const int rbuf = (s + 2) % 3, wbuf = s % 3;  // read what stage s-1 wrote, write this stage's
    // ---- consume: wave w takes k-blocks 4w..4w+3 of every row tile
    float acc = 0.0f;
    if (s > 0) {
      if (VARIANT == 1 && CANARY) {
        // wait on ONE unit (the last tile of this wave's last k-block) before the bulk fetch, so that
        // the bulk loads rarely come back with the reset pattern
        const u32x4* p = reinterpret_cast<const u32x4*>(base + unit(rbuf, tiles, TILES - 1, 4 * w + 3)) + lane;
        u32x4 x = load_sc1(p);
        while (__any(x[0] == SENT || x[1] == SENT || x[2] == SENT || x[3] == SENT)) {
          if (++spins > (1u << 22)) break;
          x = load_sc1(p);
        }
      }
      // all loads in flight first (as the kernel's operand fetch does), then look at them
      u32x4 v[TILES][4];
#pragma unroll
      for (int tile = 0; tile < TILES; ++tile)
#pragma unroll
        for (int k = 0; k < 4; ++k)
          v[tile][k] = issue_sc1(reinterpret_cast<const u32x4*>(base + unit(rbuf, tiles, tile, 4 * w + k)) + lane);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int tile = 0; tile < TILES; ++tile)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          u32x4 x = v[tile][k];
          asm volatile("" : "+v"(x));
          if (VARIANT == 1) {
            const u32x4* p = reinterpret_cast<const u32x4*>(base + unit(rbuf, tiles, tile, 4 * w + k)) + lane;
            while (__any(x[0] == SENT || x[1] == SENT || x[2] == SENT || x[3] == SENT)) {
              if (++spins > (1u << 22)) break;
              x = load_sc1(p);
            }
          }
          acc += __uint_as_float(x[0]) + __uint_as_float(x[1]) + __uint_as_float(x[2]) + __uint_as_float(x[3]);
        }
    }
    // fold to one number per workgroup (stands in for the split-K combine)
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) s_part[w] = acc;
    __syncthreads();
    float sum = 0.0f;
    for (int i = 0; i < 8; ++i) sum += s_part[i];
    if (VARIANT == 1 && s > 0) {
      // everybody's stage s-1 output has been read by this workgroup; what THIS workgroup wrote at
      // stage s-2 has been read by everybody who produced stage s-1 output, i.e. by all: reset it
      // (issued before the stage's arithmetic, acknowledged by the time that is done)
      const int zbuf = (s + 1) % 3;
      for (int e = t; e < tiles * 64; e += 512) {
        u32x4* p = reinterpret_cast<u32x4*>(base + unit(zbuf, tiles, e >> 6, rank)) + (e & 63);
        *p = u32x4{SENT, SENT, SENT, SENT};
      }
    }
    for (int i = 0; i < busy; ++i) sum = sum * 0.999f + 0.001f;  // stand-in for the stage's arithmetic
    carry = sum * (1.0f / 65536.0f) + (float)rank * 0.001f + 0.25f;
    if (VARIANT == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the reset is in L2 before any of this stage's data
    // ---- produce: rows x 16 values
    for (int e = t; e < tiles * 64; e += 512) {
      f32x4* p = reinterpret_cast<f32x4*>(base + unit(wbuf, tiles, e >> 6, rank)) + (e & 63);
      const float v = carry + (float)(e & 63) * 1e-6f;
      *p = f32x4{v, v + 1e-3f, v + 2e-3f, v + 3e-3f};
    }
    if (VARIANT == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t == 0) {
        (void)__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t target = 32u * (uint32_t)(s + 1);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1u << 22)) break;
        }
      }
      __syncthreads();
    } else {
      __syncthreads();  // s_part is reused
    }
  }
  if (t == 0) { out[blockIdx.x] = carry; spins_out[blockIdx.x] = spins; }
}

int main(int argc, char** argv) {
  const int tiles = 5;  // (the kernel's register arrays are sized for it)
  (void)argv;
  const int stages = argc > 1 ? atoi(argv[1]) : 6000;
  const int busy = argc > 2 ? atoi(argv[2]) : 0;
  float* d_data; uint32_t* d_bar; float* d_out; uint32_t* d_spins;
  const size_t words = (size_t)NCL * 3 * tiles * KB * 256;
  HIPCHK(hipMalloc(&d_data, words * 4));
  HIPCHK(hipMalloc(&d_bar, NCL * 32 * 4));
  HIPCHK(hipMalloc(&d_out, 256 * 4));
  HIPCHK(hipMalloc(&d_spins, 256 * 4));
  hipEvent_t a, b;
  HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
  std::vector<float> res[3];
  for (int rep = 0; rep < 2; ++rep)
    for (int variant = 0; variant < 3; ++variant) {
      std::vector<uint32_t> init(words, SENT);
      HIPCHK(hipMemcpy(d_data, init.data(), words * 4, hipMemcpyHostToDevice));
      HIPCHK(hipMemset(d_bar, 0, NCL * 32 * 4));
      HIPCHK(hipEventRecord(a));
      if (variant == 0) hipLaunchKernelGGL((k_handoff<0, 5, false>), dim3(256), dim3(512), 0, 0, d_data, d_bar, tiles, stages, busy, d_out, d_spins);
      else if (variant == 1) hipLaunchKernelGGL((k_handoff<1, 5, false>), dim3(256), dim3(512), 0, 0, d_data, d_bar, tiles, stages, busy, d_out, d_spins);
      else hipLaunchKernelGGL((k_handoff<1, 5, true>), dim3(256), dim3(512), 0, 0, d_data, d_bar, tiles, stages, busy, d_out, d_spins);
      HIPCHK(hipGetLastError());
      HIPCHK(hipEventRecord(b));
      HIPCHK(hipEventSynchronize(b));
      float ms = 0;
      HIPCHK(hipEventElapsedTime(&ms, a, b));
      std::vector<float> out(256);
      std::vector<uint32_t> spins(256);
      HIPCHK(hipMemcpy(out.data(), d_out, 256 * 4, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(spins.data(), d_spins, 256 * 4, hipMemcpyDeviceToHost));
      uint64_t sp = 0; uint32_t spmax = 0;
      for (uint32_t v : spins) { sp += v; spmax = v > spmax ? v : spmax; }
      printf("variant %s rep %d: %.3f us per stage (tiles %d, busy %d), out[0]=%.9g out[255]=%.9g, polls/stage/wg %.1f max %u\n",
             variant == 0 ? "A barrier " : variant == 1 ? "B in-band " : "B2 canary ", rep, ms * 1e3 / stages, tiles, busy, out[0], out[255],
             (double)sp / 256.0 / stages, spmax);
      res[variant] = out;
    }
  int same = 1;
  for (int i = 0; i < 256; ++i) same &= res[0][i] == res[1][i] && res[0][i] == res[2][i];
  printf("results identical: %s\n", same ? "yes" : "NO");
  return same ? 0 : 1;
}
