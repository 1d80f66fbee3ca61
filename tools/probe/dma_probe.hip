// Probe: global_load_lds (LDS-DMA) 16-byte form: layout, per-lane source address, __syncthreads visibility.
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ void dma16(const float* gsrc, float* lds_dst) {
  __builtin_amdgcn_global_load_lds(GLB_PTR(gsrc), LDS_PTR(lds_dst), 16, 0, 0);
}

// src: rows of `ld` floats; copy nrows x 128 floats into lds [nrows][128]; then dump lds to out
extern "C" __global__ void probe_dma(const float* src, int ld, int nrows, float* out) {
  __shared__ float lds[32 * 128];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // one wave-op = 2 rows x 512 B; nrows/2 ops spread over 4 waves
  for (int op = wave; op < nrows / 2; op += 4) {
    const int row = 2 * op + (lane >> 5);
    dma16(src + (size_t)row * ld + (lane & 31) * 4, lds + op * 256);   // wave-uniform LDS base, HW adds lane*16
  }
  __syncthreads();
  for (int i = tid; i < nrows * 128; i += 256) out[i] = lds[i];
}

extern "C" int probe_dma_run(const float* src, int ld, int nrows, float* out, void* stream) {
  hipLaunchKernelGGL(probe_dma, dim3(1), dim3(256), 0, (hipStream_t)stream, src, ld, nrows, out);
  return (int)hipGetLastError();
}
