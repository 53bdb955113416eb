// Developer tool: achievable HBM rates on this MI355X for the access shapes of the tile
// kernel (streaming loads, 16-byte streaming stores, copy).  hipcc --offload-arch=gfx950 -O3
// tools/membench.hip -o gpurun_out/membench && gpurun_out/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double2_ __attribute__((ext_vector_type(2)));
__global__ void k_store(double2_* out, size_t n2) {
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (; i < n2; i += stride) out[i] = double2_{double(i), 1.0};
}
__global__ void k_load(const double2_* in, double* sink, size_t n2) {
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  double acc = 0;
  for (; i < n2; i += stride) { double2_ v = in[i]; acc += v.x + v.y; }
  if (acc == 1.2345e300) sink[0] = acc;
}
__global__ void k_copy(const double2_* in, double2_* out, size_t n2) {
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (; i < n2; i += stride) out[i] = in[i];
}
// one block writes a contiguous 28 KB tile chunk (like S5), blocks strided over the array
__global__ void k_store_tiles(double2_* out, size_t n2, int per_tile2) {
  for (size_t t = blockIdx.x; t * per_tile2 < n2; t += gridDim.x) {
    double2_* o = out + t * per_tile2;
    for (int i = threadIdx.x; i < per_tile2 && t * per_tile2 + i < n2; i += blockDim.x) o[i] = double2_{1.0, 2.0};
  }
}
// same, but every 16-byte store starts 8 bytes off a 16-byte boundary (as when a CSR run
// starts at an odd entry), and an 8-byte-per-lane variant
__global__ void k_store_tiles_misaligned(double* out, size_t n, int per_tile) {
  for (size_t t = blockIdx.x; (t + 1) * per_tile + 1 < n; t += gridDim.x) {
    double* o = out + t * per_tile + 1;
    for (int i = 2 * threadIdx.x; i + 1 < per_tile; i += 2 * blockDim.x)
      *reinterpret_cast<double2_*>(o + i) = double2_{1.0, 2.0};
  }
}
__global__ void k_store_tiles_8b(double* out, size_t n, int per_tile) {
  for (size_t t = blockIdx.x; (t + 1) * per_tile + 1 < n; t += gridDim.x) {
    double* o = out + t * per_tile + 1;
    for (int i = threadIdx.x; i < per_tile; i += blockDim.x) o[i] = 3.0;
  }
}
template <typename F> float timeit(F f, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); for (int r = 0; r < reps; ++r) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
  const size_t bytes = size_t(280) << 20, n2 = bytes / 16;
  double2_ *a, *b; double* sink;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&sink, 8);
  hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
  for (int blocks : {512, 1024, 2048, 8192}) {
    float s = timeit([&] { hipLaunchKernelGGL(k_store, dim3(blocks), dim3(256), 0, 0, a, n2); }, 20);
    float l = timeit([&] { hipLaunchKernelGGL(k_load, dim3(blocks), dim3(256), 0, 0, a, sink, n2); }, 20);
    float c = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, b, n2); }, 20);
    float t = timeit([&] { hipLaunchKernelGGL(k_store_tiles, dim3(blocks), dim3(512), 0, 0, a, n2, 1792); }, 20);
    float m = timeit([&] { hipLaunchKernelGGL(k_store_tiles_misaligned, dim3(blocks), dim3(512), 0, 0, (double*)a, bytes / 8, 3584); }, 20);
    float e = timeit([&] { hipLaunchKernelGGL(k_store_tiles_8b, dim3(blocks), dim3(512), 0, 0, (double*)a, bytes / 8, 3584); }, 20);
    printf("blocks %5d: misaligned 16B tile-store %.1f us (%.2f TB/s)   8B tile-store %.1f us (%.2f TB/s)\n", blocks,
           m * 1e3, bytes / m / 1e9, e * 1e3, bytes / e / 1e9);
    printf("blocks %5d: store %.1f us (%.2f TB/s)  load %.1f us (%.2f TB/s)  copy %.1f us (%.2f TB/s rd+wr)  tile-store %.1f us (%.2f TB/s)\n",
           blocks, s * 1e3, bytes / s / 1e9, l * 1e3, bytes / l / 1e9, c * 1e3, 2.0 * bytes / c / 1e9, t * 1e3, bytes / t / 1e9);
  }
  return 0;
}
