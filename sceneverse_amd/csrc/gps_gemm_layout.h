// gps_gemm_layout.h -- LDS image and fragment index maps of the bf16 MFMA GEMM (gps_gemm.hip).
//
// Kept free of HIP constructs so that the SAME functions are compiled for the device (hipcc) and for the host
// emulation that checks them end to end on a CPU (tests/test_gemm_layout.py builds tools/gemm_layout_shim.cpp
// with g++): stage map -> LDS image -> fragment reads -> MFMA lane semantics -> C = op(A) op(B).
//
// One K step of a workgroup tile stages BK = 64 reduction indices of both operands into LDS with
// global_load_lds_dwordx4: a wave-instruction ("piece") moves 64 x 16 B to  piece_base + lane * 16  (the LDS
// image is lane-linear, MI355X guide section 5), the per-lane SOURCE address is free.  All swizzles are
// therefore permutations of the source chunks inside one row, applied again on the read side.
//
//  K-major operand ("KM": rows = m or n, 64 k contiguous, 128 B per row, 8 chunks of 16 B):
//      position (16-B units) = row * 8 + (chunk ^ ((row >> 1) & 7))
//      fragment: lane (i, g) reads k = 32 ks + 8 g .. + 7 of row r0 + i with one ds_read_b128; the XOR spreads the
//      16 lanes of every b128 lane group over all 16 slots of a 256-B bank row.
//  reduction-major operand ("RM": rows = k, COLS = tile width of m or n contiguous, COLS / 8 chunks per row),
//      read with ds_read_b64_tr_b16 (each 16-lane group fetches a 4 (k) x 16 (cols) block and gets it transposed:
//      lane i of the group ends up with the 4 k values of column i):
//      position = k * (COLS / 8) + (chunk ^ 2 * ((k & 3) | ((k >> 1) & 4)))      (COLS == 64: 2 * (k & 3))
//      group g fetches k rows 32 ks + 8 g + {0..3} (first read) and + 4 (second read): the two groups of a half-wave
//      touch rows {0..3, 8..11} (then {4..7, 12..15}), whose swizzle keys (k bits 0, 1, 3) are 8 different values
//      -> 8 different 32-byte column pairs of the 256-B bank row.
//  Both fragment kinds hand lane (i, g) the k values 32 ks + 8 g + e, e = 0..7 -- the MFMA's own operand order.
#ifndef GPS_GEMM_LAYOUT_H_
#define GPS_GEMM_LAYOUT_H_

#if defined(__HIPCC__)
#define GPS_HD __host__ __device__ __forceinline__
#else
#define GPS_HD inline
#endif

namespace gps_gemm_layout {

constexpr int BK = 64;          // reduction indices per stage
constexpr int PIECE = 1024;     // bytes one wave-instruction of global_load_lds_dwordx4 moves

// ---- K-major operand ------------------------------------------------------------------------------------
// piece q (0 .. rows/8 - 1) of a [rows][64] tile: which (row, source chunk) lane `lane` fetches
GPS_HD void km_stage_src(int q, int lane, int &row, int &chunk) {
  row = 8 * q + (lane >> 3);
  chunk = (lane & 7) ^ ((row >> 1) & 7);
}
// byte offset of (row, 16-byte chunk c) inside the tile image
GPS_HD int km_chunk_off(int row, int c) { return row * 128 + ((c ^ ((row >> 1) & 7)) << 4); }
// fragment: one 16-byte read
GPS_HD int km_frag(int row, int ks, int g) { return km_chunk_off(row, 4 * ks + g); }

// ---- reduction-major operand ----------------------------------------------------------------------------
template <int COLS>
GPS_HD int rm_swz(int k) { return COLS >= 128 ? 2 * ((k & 3) | ((k >> 1) & 4)) : 2 * (k & 3); }
template <int COLS>
GPS_HD void rm_stage_src(int q, int lane, int &k, int &chunk) {
  constexpr int CPR = COLS / 8;                          // chunks per row
  const int p = 64 * q + lane;
  k = p / CPR;
  chunk = (p % CPR) ^ rm_swz<COLS>(k);
}
// byte address a lane hands to ds_read_b64_tr_b16: lane = 16 g + i, fragment columns col0 .. col0 + 15
template <int COLS>
GPS_HD int rm_frag(int col0, int ks, int lane, int which) {
  const int i = lane & 15, g = lane >> 4;
  const int k = 32 * ks + 8 * g + 4 * which + (i >> 2);
  const int col = col0 + 4 * (i & 3);
  return k * (COLS * 2) + (((col >> 3) ^ rm_swz<COLS>(k)) << 4) + (((col >> 2) & 1) << 3);
}

// k index (inside the 32-wide MFMA K step) that fragment element e (0..7) of lane group g stands for
GPS_HD int frag_k(int g, int e) { return 8 * g + e; }

// XCD-aware block id -> virtual id: the blocks that land on one XCD (block id mod 8, observed dispatch rule,
// used for speed only) get a CONTIGUOUS range of virtual ids; bijective for every total
GPS_HD int xcd_virtual_id(int bid, int total) {
  const int xcd = bid & 7, slot = bid >> 3;
  const int q = total >> 3, r = total & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

}  // namespace gps_gemm_layout
#endif  // GPS_GEMM_LAYOUT_H_
