// Probe: what does a tiny dependent kernel cost on MI355X, by grid size and by the number of
// dependent global round trips inside it?  Timed with hipExtLaunchKernelGGL start/stop events
// (dispatch begin/end timestamps) and by wall clock over a chain of launches.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_empty(int* p) { if (p == nullptr) *p = 0; }
__global__ void k_read_exit(const int* p) { if (p[0] > 1000000) ((int*)p)[1] = 1; }
// chase: `hops` dependent loads, then one store
__global__ void k_chase(const int* next, int* out, int hops) {
  int i = blockIdx.x * 64 + (threadIdx.x & 63);
  for (int h = 0; h < hops; ++h) i = next[i];
  if (threadIdx.x == 0) out[blockIdx.x] = i;
}
// writer: touch the chase table so the next kernel finds it dirty / elsewhere
__global__ void k_touch(int* next, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) next[i] = next[i];
}

template <typename F>
double time_chain(hipStream_t s, int reps, F launch, double* ev_avg_us) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 20; ++i) launch(nullptr, nullptr);
  hipStreamSynchronize(s);
  double ev = 0;
  for (int i = 0; i < 50; ++i) { launch(a, b); hipStreamSynchronize(s); float ms; hipEventElapsedTime(&ms, a, b); ev += ms * 1e3; }
  *ev_avg_us = ev / 50;
  auto t0 = std::chrono::high_resolution_clock::now();
  for (int i = 0; i < reps; ++i) launch(nullptr, nullptr);
  hipStreamSynchronize(s);
  auto t1 = std::chrono::high_resolution_clock::now();
  return std::chrono::duration<double, std::micro>(t1 - t0).count() / reps;
}

int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const int n = 1 << 20;
  int *d_next, *d_out; CK(hipMalloc(&d_next, n * 4)); CK(hipMalloc(&d_out, 1 << 16));
  std::vector<int> h(n);
  for (int i = 0; i < n; ++i) h[i] = (int)(((long long)i * 7919 + 12345) % n);
  CK(hipMemcpy(d_next, h.data(), n * 4, hipMemcpyHostToDevice));
  double ev;
  for (int blocks : {1, 64, 704, 1280, 5120}) {
    for (int threads : {256, 512}) {
      double w = time_chain(s, 2000, [&](hipEvent_t a, hipEvent_t b) {
        if (a) hipExtLaunchKernelGGL(k_empty, dim3(blocks), dim3(threads), 0, s, a, b, 0, d_out);
        else hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(threads), 0, s, d_out);
      }, &ev);
      printf("empty      blocks %5d x %3d : chain %.2f us/launch, event %.2f us\n", blocks, threads, w, ev);
    }
  }
  for (int blocks : {64, 1280}) {
    double w = time_chain(s, 2000, [&](hipEvent_t a, hipEvent_t b) {
      if (a) hipExtLaunchKernelGGL(k_read_exit, dim3(blocks), dim3(512), 0, s, a, b, 0, d_next);
      else hipLaunchKernelGGL(k_read_exit, dim3(blocks), dim3(512), 0, s, d_next);
    }, &ev);
    printf("read+exit  blocks %5d x 512 : chain %.2f us/launch, event %.2f us\n", blocks, w, ev);
  }
  for (int hops : {1, 2, 3, 4, 8}) {
    double w = time_chain(s, 2000, [&](hipEvent_t a, hipEvent_t b) {
      if (a) hipExtLaunchKernelGGL(k_chase, dim3(64), dim3(256), 0, s, a, b, 0, d_next, d_out, hops);
      else hipLaunchKernelGGL(k_chase, dim3(64), dim3(256), 0, s, d_next, d_out, hops);
    }, &ev);
    printf("chase %d hops (same table, warm) : chain %.2f us/launch, event %.2f us\n", hops, w, ev);
  }
  // cold chase: a writer kernel touches the table between launches (as the decode's producers do)
  for (int hops : {1, 2, 4}) {
    double w = time_chain(s, 1000, [&](hipEvent_t a, hipEvent_t b) {
      hipLaunchKernelGGL(k_touch, dim3(n / 256), dim3(256), 0, s, d_next, n);
      if (a) hipExtLaunchKernelGGL(k_chase, dim3(64), dim3(256), 0, s, a, b, 0, d_next, d_out, hops);
      else hipLaunchKernelGGL(k_chase, dim3(64), dim3(256), 0, s, d_next, d_out, hops);
    }, &ev);
    printf("chase %d hops after a 4 MB writer : pair %.2f us, chase event %.2f us\n", hops, w, ev);
  }
  return 0;
}
