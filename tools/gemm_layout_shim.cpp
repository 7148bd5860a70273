// C shim over sceneverse_amd/csrc/gps_gemm_layout.h for the host-side layout emulation
// (tests/test_gemm_layout.py compiles it with g++ and drives it through ctypes).
#include "gps_gemm_layout.h"
using namespace gps_gemm_layout;
extern "C" {
void shim_km_stage_src(int q, int lane, int *row, int *chunk) { km_stage_src(q, lane, *row, *chunk); }
int shim_km_frag(int row, int ks, int g) { return km_frag(row, ks, g); }
void shim_rm_stage_src(int cols, int q, int lane, int *k, int *chunk) {
  if (cols == 64) rm_stage_src<64>(q, lane, *k, *chunk);
  else if (cols == 128) rm_stage_src<128>(q, lane, *k, *chunk);
  else rm_stage_src<256>(q, lane, *k, *chunk);
}
int shim_rm_frag(int cols, int col0, int ks, int lane, int which) {
  if (cols == 64) return rm_frag<64>(col0, ks, lane, which);
  if (cols == 128) return rm_frag<128>(col0, ks, lane, which);
  return rm_frag<256>(col0, ks, lane, which);
}
int shim_frag_k(int g, int e) { return frag_k(g, e); }
int shim_xcd_virtual_id(int bid, int total) { return xcd_virtual_id(bid, total); }
}
