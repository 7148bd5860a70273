// Probe for gfx950's LDS transpose read `ds_read_b64_tr_b16` (groundwork for the split-K weight-gradient
// GEMM of DESIGN.md section 9: token-major tiles stored as loaded and read transposed).  LDS slot e (bf16
// sized) holds the integer e, so results can be read back exactly; every lane issues ONE transpose read at a
// caller-given byte offset into LDS and the four 16-bit values it receives are written out:
//   out[lane][0..3] = the slot numbers the lane received (element order = register order).
// Build + run on a GPU box: python tools/probes/run_tr_b16_probe.py
#include <hip/hip_runtime.h>
#include <stdint.h>

extern "C" __global__ void tr_b16_probe(const int *__restrict__ lane_byte_offset, uint16_t *__restrict__ out,
                                        int n_slots) {
  extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
  for (int e = threadIdx.x; e < n_slots; e += blockDim.x) lds[e] = (uint16_t)e;
  __syncthreads();
  const unsigned addr = (unsigned)(uintptr_t)lds + (unsigned)lane_byte_offset[threadIdx.x];
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[threadIdx.x * 4 + 0] = (uint16_t)(v & 0xFFFFull);
  out[threadIdx.x * 4 + 1] = (uint16_t)((v >> 16) & 0xFFFFull);
  out[threadIdx.x * 4 + 2] = (uint16_t)((v >> 32) & 0xFFFFull);
  out[threadIdx.x * 4 + 3] = (uint16_t)(v >> 48);
}
