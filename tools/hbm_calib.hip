// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access width the extraction kernels use: one aligned
// dword per lane, consecutive lanes on consecutive dwords (the guide calibrates only 16 B/lane).  The buffer is 1 GiB,
// four times the Infinity Cache, so every byte comes from / goes to HBM.  Known traffic: k_calib_read reads BYTES,
// k_calib_write writes BYTES.  Build: hipcc --offload-arch=gfx950 -O3 tools/hbm_calib.hip -o tools/hbm_calib.bin
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr size_t BYTES = 1ull << 30;

__global__ __launch_bounds__(256) void k_calib_read(const unsigned *__restrict__ src, unsigned *__restrict__ sink, size_t n) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc ^= src[i];
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_calib_write(unsigned *__restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = (unsigned)i;
}

int main() {
  unsigned *buf, *sink;
  if (hipMalloc(&buf, BYTES) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { fprintf(stderr, "alloc failed\n"); return 1; }
  hipMemset(buf, 1, BYTES);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(k_calib_read, dim3(8192), dim3(256), 0, 0, buf, sink, BYTES / 4);
    hipLaunchKernelGGL(k_calib_write, dim3(8192), dim3(256), 0, 0, buf, BYTES / 4);
  }
  hipDeviceSynchronize();
  printf("known bytes per launch: %zu\n", BYTES);
  return 0;
}
