// Semantics probe for gfx950's LDS transpose read (ds_read_b64_tr_b16), GPU box only:
//   hipcc --offload-arch=gfx950 -O2 tools/tr_probe.hip -o tools/tr_probe && tools/tr_probe
// LDS is filled with u16 values equal to their element index; every lane issues one
// ds_read_b64_tr_b16 at a probe address and the four returned elements are printed per lane.
// Probe 0: lane l -> byte address 8*l (the packed [4][16] block per 16-lane group).
// Probe 1: lane l -> row (l>>4)*8 + ((l&15)>>2) of a 256-B-row image, column 4*(l&3)
//          (the address pattern the k-major GEMM loaders use).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef short v4i16 __attribute__((ext_vector_type(4)));

__global__ void probe(uint16_t* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x;
  int byte;
  if (mode == 0) byte = 8 * l;
  else byte = (((l >> 4) * 8 + ((l & 15) >> 2)) * 256) + 8 * (l & 3);
  v4i16 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4i16 __attribute__((address_space(3)))*)((__attribute__((address_space(3))) char*)lds + byte));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}

int main() {
  uint16_t* d;
  uint16_t h[256];
  hipMalloc(&d, sizeof(h));
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d:", l);
      for (int j = 0; j < 4; ++j) {
        if (mode == 0) printf(" %4d", h[l * 4 + j]);
        else printf(" (r%2d,c%2d)", h[l * 4 + j] / 128, h[l * 4 + j] % 128);
      }
      printf("\n");
    }
  }
  return 0;
}
