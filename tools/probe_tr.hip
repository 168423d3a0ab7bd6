// Probe of ds_read_b64_tr_b16 semantics on gfx950: LDS holds u16 value = element index; every lane passes its own
// byte address; we print which element each lane's 4 results came from.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(const int* addr_bytes, uint16_t* out) {
  __shared__ uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  unsigned a = (unsigned)(uintptr_t)lds + addr_bytes[threadIdx.x];
  uint2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
  out[threadIdx.x * 4 + 0] = r.x & 0xffff; out[threadIdx.x * 4 + 1] = r.x >> 16;
  out[threadIdx.x * 4 + 2] = r.y & 0xffff; out[threadIdx.x * 4 + 3] = r.y >> 16;
}
int main() {
  int h[64]; uint16_t ho[256]; int* d; uint16_t* o;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
  for (int pat = 0; pat < 3; ++pat) {
    for (int l = 0; l < 64; ++l) {
      if (pat == 0) h[l] = l * 8;                                  // contiguous: lane l -> elements 4l..4l+3
      if (pat == 1) h[l] = ((l & 15) >> 2) * 64 + (l & 3) * 8 + (l >> 4) * 512;  // 4 rows x 16 cols, pitch 32 el, per 16-lane group
      if (pat == 2) h[l] = (l & 15) * 200 + (l >> 4) * 8;          // every lane its own row (pitch 100 el)
    }
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d, o);
    hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    printf("pattern %d (lane: addr_elem -> 4 results as element indices)\n", pat);
    for (int l = 0; l < 64; ++l)
      printf("  lane %2d addr_el %4d -> %4d %4d %4d %4d\n", l, h[l] / 2, ho[l * 4], ho[l * 4 + 1], ho[l * 4 + 2], ho[l * 4 + 3]);
  }
  return 0;
}
