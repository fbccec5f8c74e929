// Batched strided fp32 GEMM on the CDNA4 f32 matrix core (v_mfma_f32_32x32x2_f32), with an optional
// fused RBF epilogue.  One 256-thread workgroup (4 wave64, 2x2) computes a BM x BN tile; K is walked
// in slabs of 16 staged through LDS with a register prefetch of the next slab.
//
// LDS images mirror the global layout of each operand so that staging is one 16-byte write per float4 and the MFMA
// fragment reads are bank-conflict free:
//   K-contiguous operand  -> [row][BK], unpadded, 16-byte chunk c of row r stored at chunk c ^ f(r) (XOR swizzle):
//                            a lane reads its row's 4 consecutive k with ONE ds_read_b128 (the 16 lanes of a
//                            b128 group have distinct r & 15, hence distinct slots), staging writes are b128;
//   M/N-contiguous operand-> [k][rows+4] (lanes read consecutive floats, b128 staging writes).
// MFMA 32x32x2 fragment maps (cdna guide §3): lane l holds A[i=l&31][k=l>>5], B[k=l>>5][j=l&31];
// D register r of lane l is row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31.
// The k index an MFMA sums over is free as long as A and B agree: within a group of 8 k the half-wave h = l>>5
// owns k = 8g + 4h + j, j = 0..3, i.e. one float4 per lane feeds four MFMAs.
// On gfx950 the f32 MFMA and the VALU/LDS/SALU issue of a SIMD do not overlap (measured: a slab costs its MFMA
// cycles PLUS the issue cycles of everything else), so the kernel is written to minimise non-MFMA instructions.
#include "common.h"
#include "chol_small3.h"
#include "t0_bwd_mat.h"
#include "t0_prologue.h"
#include <atomic>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

STEP_SPAN_TABLE(gemm)
#ifdef STEP_SPANS      // start / end of every matrix chain of the backward's merged launch
__device__ unsigned long long g_bmat_ends[64][2];
extern "C" void vargp_debug_bmat_ends(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bmat_ends), sizeof(g_bmat_ends)); }
#endif

namespace vargp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Where the out-of-range lanes of an edge tile's epilogue store: with `if (row < M && col < N) C[..] = v` every element is a
// branch, and the compiler drains vmcnt before each (one memory round trip per element: 32-64 of them per thread, most of a
// small product's duration).  A select between the real address and this dump keeps the epilogue straight-line.  Never read.
__device__ float g_gemm_trash[512];

template <bool KC, int ROWS, int BK>
struct LdsLayout {
  static constexpr int kStride = KC ? BK : (ROWS + 4);
  static constexpr int kSize = KC ? ROWS * BK : BK * (ROWS + 4);
  __device__ static __forceinline__ int at(int r, int k) {
    // chunk swizzle: rows that share a 256-byte bank row (64/BK of them) keep their chunk order, the next group of
    // rows is rotated by one more: 16 lanes with distinct (row & 15) always hit 16 distinct 16-byte slots
    if constexpr (KC) return r * BK + ((((k >> 2) ^ (r / (64 / BK))) & (BK / 4 - 1)) << 2) + (k & 3);
    else return k * kStride + r;
  }
};

// Global -> registers for one ROWS x BK slab.  Element (r,k) lives at base[(r0+r)*ld + k] (KC) or
// base[k*ld + r0 + r] (!KC).  Out-of-range elements (r >= rmax, k >= ke) read as zero.
template <bool KC, int ROWS, int BK, bool VEC, int NT = 256>
__device__ __forceinline__ void load_slab(const float* __restrict__ base, int ld, int r0, int rmax, int k0,
                                          int ke, float (&reg)[ROWS * BK / NT]) {
  constexpr int NPT = ROWS * BK / NT;
  const int tid = threadIdx.x;
  if constexpr (VEC) {
#pragma unroll
    for (int c = 0; c < NPT / 4; ++c) {
      const int q = tid + NT * c;
      int r, k;
      if constexpr (KC) { r = q / (BK / 4); k = (q % (BK / 4)) * 4; } else { k = q / (ROWS / 4); r = (q % (ROWS / 4)) * 4; }
      const int gr = r0 + r, gk = k0 + k;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (KC) {
        if (gr < rmax) {
          const float* src = base + (int64_t)gr * ld + gk;
          if (gk + 3 < ke) v = *reinterpret_cast<const float4*>(src);
          else {
            if (gk < ke) v.x = src[0];
            if (gk + 1 < ke) v.y = src[1];
            if (gk + 2 < ke) v.z = src[2];
          }
        }
      } else {
        if (gk < ke) {
          const float* src = base + (int64_t)gk * ld + gr;
          if (gr + 3 < rmax) v = *reinterpret_cast<const float4*>(src);
          else {
            if (gr < rmax) v.x = src[0];
            if (gr + 1 < rmax) v.y = src[1];
            if (gr + 2 < rmax) v.z = src[2];
          }
        }
      }
      reg[4 * c + 0] = v.x; reg[4 * c + 1] = v.y; reg[4 * c + 2] = v.z; reg[4 * c + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int c = 0; c < NPT; ++c) {
      const int e = tid + NT * c;
      int r, k;
      if constexpr (KC) { r = e / BK; k = e % BK; } else { k = e / ROWS; r = e % ROWS; }
      const int gr = r0 + r, gk = k0 + k;
      float v = 0.f;
      if (gr < rmax && gk < ke) v = KC ? base[(int64_t)gr * ld + gk] : base[(int64_t)gk * ld + gr];
      reg[c] = v;
    }
  }
}

// per-k scale factors of the elements this thread stages (K-contiguous operand only): with VEC the
// thread owns k = 4*(tid % (BK/4)) + e, e<4, in every chunk (256 % (BK/4) == 0); without, k = tid % BK.
// Loaded together with the slab (prefetch), applied when the slab is written to LDS.
template <int BK, bool VEC>
__device__ __forceinline__ void load_scale(const float* __restrict__ kscale, int k0, int ke, float (&rs)[4]) {
  const int tid = threadIdx.x;
  if constexpr (VEC) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = k0 + (tid % (BK / 4)) * 4 + e;
      rs[e] = (k < ke) ? kscale[k] : 0.f;
    }
  } else {
    const int k = k0 + (tid % BK);
    rs[0] = (k < ke) ? kscale[k] : 0.f;
  }
}

// registers -> LDS (same element <-> thread map as load_slab); optional per-k scale (KC only)
template <bool KC, int ROWS, int BK, bool VEC, bool SCALE, int NT = 256>
__device__ __forceinline__ void store_slab(float* __restrict__ lds, const float (&reg)[ROWS * BK / NT],
                                           const float (&rs)[4]) {
  constexpr int NPT = ROWS * BK / NT;
  using L = LdsLayout<KC, ROWS, BK>;
  const int tid = threadIdx.x;
  if constexpr (VEC) {
#pragma unroll
    for (int c = 0; c < NPT / 4; ++c) {
      const int q = tid + NT * c;
      if constexpr (KC) {
        const int r = q / (BK / 4), k = (q % (BK / 4)) * 4;
        float4 v = make_float4(reg[4 * c], reg[4 * c + 1], reg[4 * c + 2], reg[4 * c + 3]);
        if constexpr (SCALE) { v.x *= rs[0]; v.y *= rs[1]; v.z *= rs[2]; v.w *= rs[3]; }
        *reinterpret_cast<float4*>(&lds[L::at(r, k)]) = v;
      } else {
        const int k = q / (ROWS / 4), r = (q % (ROWS / 4)) * 4;
        *reinterpret_cast<float4*>(&lds[L::at(r, k)]) =
            make_float4(reg[4 * c], reg[4 * c + 1], reg[4 * c + 2], reg[4 * c + 3]);
      }
    }
  } else {
#pragma unroll
    for (int c = 0; c < NPT; ++c) {
      const int e = tid + NT * c;
      int r, k;
      if constexpr (KC) { r = e / BK; k = e % BK; } else { k = e / ROWS; r = e % ROWS; }
      float v = reg[c];
      if constexpr (SCALE) v *= rs[0];
      lds[L::at(r, k)] = v;
    }
  }
}

// ---- the same staging, one piece at a time (a piece = one float4 chunk with VEC, one float without),
// so that the main loop can drop pieces into the shadow of individual MFMAs ----------------------------
template <bool KC, int ROWS, int BK, bool VEC, int NT = 256>
struct Pieces { static constexpr int kCount = VEC ? ROWS * BK / NT / 4 : ROWS * BK / NT; };

template <bool KC, int ROWS, int BK, bool VEC, int NT = 256>
__device__ __forceinline__ void load_piece(const float* __restrict__ base, int ld, int r0, int rmax, int k0, int ke,
                                           int c, float (&reg)[ROWS * BK / NT]) {
  const int tid = threadIdx.x;
  if constexpr (VEC) {
    const int q = tid + NT * c;
    int r, k;
    if constexpr (KC) { r = q / (BK / 4); k = (q % (BK / 4)) * 4; } else { k = q / (ROWS / 4); r = (q % (ROWS / 4)) * 4; }
    const int gr = r0 + r, gk = k0 + k;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (KC) {
      if (gr < rmax) {
        const float* src = base + (int64_t)gr * ld + gk;
        if (gk + 3 < ke) v = *reinterpret_cast<const float4*>(src);
        else { if (gk < ke) v.x = src[0]; if (gk + 1 < ke) v.y = src[1]; if (gk + 2 < ke) v.z = src[2]; }
      }
    } else {
      if (gk < ke) {
        const float* src = base + (int64_t)gk * ld + gr;
        if (gr + 3 < rmax) v = *reinterpret_cast<const float4*>(src);
        else { if (gr < rmax) v.x = src[0]; if (gr + 1 < rmax) v.y = src[1]; if (gr + 2 < rmax) v.z = src[2]; }
      }
    }
    reg[4 * c + 0] = v.x; reg[4 * c + 1] = v.y; reg[4 * c + 2] = v.z; reg[4 * c + 3] = v.w;
  } else {
    const int e = tid + NT * c;
    int r, k;
    if constexpr (KC) { r = e / BK; k = e % BK; } else { k = e / ROWS; r = e % ROWS; }
    const int gr = r0 + r, gk = k0 + k;
    float v = 0.f;
    if (gr < rmax && gk < ke) v = KC ? base[(int64_t)gr * ld + gk] : base[(int64_t)gk * ld + gr];
    reg[c] = v;
  }
}

// Loop-invariant element offset of a VEC piece's float4 relative to the slab origin, with the row index
// clamped into the operand.  A clamped (duplicated) row/column only feeds output rows/columns >= M/N,
// which are never stored, so the fast path needs no masks and no zero-fill.
template <bool KC, int ROWS, int BK, int NT = 256>
__device__ __forceinline__ int piece_offset(int ld, int r0, int rmax, int c) {
  const int q = threadIdx.x + NT * c;
  if constexpr (KC) {
    const int r = q / (BK / 4), k = (q % (BK / 4)) * 4;
    return min(r0 + r, rmax - 1) * ld + k;
  } else {
    const int k = q / (ROWS / 4), r = (q % (ROWS / 4)) * 4;
    return k * ld + min(r0 + r, rmax - 4);       // requires rmax % 4 == 0 && rmax >= 4
  }
}
// One float4 through a buffer descriptor: per-thread byte offset `voff` is loop invariant, the slab advance is the
// scalar `soff`, so a load costs no VALU address arithmetic at all (cdna guide T8).
// (hipcc 7.2 lowers __builtin_amdgcn_raw_buffer_load_b128 to a ONE-dword load; the intrinsic is therefore declared
// directly, with the descriptor as four plain SGPR words.)
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ f32x4 llvm_amdgcn_raw_buffer_load_f32x4(i32x4 srsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ __forceinline__ i32x4 make_rsrc(const void* base, int bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  i32x4 r;
  r.x = (int)(a & 0xffffffffull);
  r.y = (int)((a >> 32) & 0xffffull);      // stride 0: raw buffer
  r.z = bytes;                              // num_records: loads past it return 0
  r.w = 0x00020000;
  return r;
}
template <int ROWS, int BK, int NT = 256>
__device__ __forceinline__ void load_piece_fast(i32x4 rsrc, int voff, int soff, int c, float (&reg)[ROWS * BK / NT]) {
  const f32x4 v = llvm_amdgcn_raw_buffer_load_f32x4(rsrc, voff, soff, 0);
  reg[4 * c + 0] = v.x; reg[4 * c + 1] = v.y; reg[4 * c + 2] = v.z; reg[4 * c + 3] = v.w;
}

template <bool KC, int ROWS, int BK, bool VEC, bool SCALE, int NT = 256>
__device__ __forceinline__ void store_piece(float* __restrict__ lds, const float (&reg)[ROWS * BK / NT],
                                            const float (&rs)[4], int c) {
  using L = LdsLayout<KC, ROWS, BK>;
  const int tid = threadIdx.x;
  if constexpr (VEC) {
    const int q = tid + NT * c;
    if constexpr (KC) {
      const int r = q / (BK / 4), k = (q % (BK / 4)) * 4;
      float4 v = make_float4(reg[4 * c], reg[4 * c + 1], reg[4 * c + 2], reg[4 * c + 3]);
      if constexpr (SCALE) { v.x *= rs[0]; v.y *= rs[1]; v.z *= rs[2]; v.w *= rs[3]; }
      *reinterpret_cast<float4*>(&lds[L::at(r, k)]) = v;
    } else {
      const int k = q / (ROWS / 4), r = (q % (ROWS / 4)) * 4;
      *reinterpret_cast<float4*>(&lds[L::at(r, k)]) =
          make_float4(reg[4 * c], reg[4 * c + 1], reg[4 * c + 2], reg[4 * c + 3]);
    }
  } else {
    const int e = tid + NT * c;
    int r, k;
    if constexpr (KC) { r = e / BK; k = e % BK; } else { k = e / ROWS; r = e % ROWS; }
    float v = reg[c];
    if constexpr (SCALE) v *= rs[0];
    lds[L::at(r, k)] = v;
  }
}

// floats of LDS gemm_body needs (two stages); the kernel owns the array so that several bodies can share one
template <int BM, int BN, int BK, bool AKC, bool BKC>
constexpr int gemm_lds_floats() {
  return 2 * (((LdsLayout<AKC, BM, BK>::kSize + 3) & ~3) + ((LdsLayout<BKC, BN, BK>::kSize + 3) & ~3));
}
constexpr int cmax(int a, int b) { return a > b ? a : b; }

// SCALED (RBF only): the K-contiguous A slab is multiplied by kscale[k] = 1/sigma_k^2 on its way into LDS.  Callers that
// hand over a pre-scaled B operand (x o w, written once per hyper-sample by the norm pass) pass kscale = NULL and get the
// unscaled instantiation: the scale loads and multiplies sit in the main loop, where nothing overlaps them with the MFMAs
// (stress K_uf tile [20480 x 784] x [8192 x 784]^T: 93 TFLOP/s scaled in the loop, 133 as a plain product).
// PIN: the fragment reads of group g + PF are pinned in front of group g's MFMAs (sched_barrier).  Left alone the scheduler sinks
// them behind the MFMAs to save eight registers -- `ds_read, s_waitcnt lgkmcnt(0), 4 MFMAs` per group, the LDS latency exposed
// every time.  Worth it where occupancy is fixed anyway or the launch is latency-bound (the merged factorisation + K_uf launch,
// the 64 x 64 x 64 tiles of the small batched products: Cfg2 step 197 -> 193.5 us, Split-MNIST t = 1 1754 -> 1794 steps/s);
// the 128-row tiles of the throughput-bound products lose occupancy to the extra live registers (S = 64: 314 -> 301 steps/s) and
// keep the compiler's schedule.
template <int BM, int BN, int BK, bool AKC, bool BKC, bool VEC, bool RBF, bool SCALED = true, int NT = 256,
          bool PIN = (BM == 64 && BN == 64)>
__device__ __forceinline__ void gemm_body(const GemmParams& p, const int tile_id_, const int batch_id_, const int split_id_,
                                          float* __restrict__ lds) {
  // The workgroup's tile / batch / split indices are wave-uniform, but they come out of integer divisions that the
  // compiler carries out on the vector unit, so it treats everything derived from them as divergent: operand base
  // pointers in VGPRs, and a waterfall loop (v_readfirstlane / v_cmp / s_and_saveexec) around EVERY buffer load of the
  // main loop, whose descriptor must be scalar.  One readfirstlane here makes all of it scalar.
  const int tile_id = __builtin_amdgcn_readfirstlane(tile_id_);
  const int batch_id = __builtin_amdgcn_readfirstlane(batch_id_);
  const int split_id = __builtin_amdgcn_readfirstlane(split_id_);
  // NT threads = NT / 64 waves in a (NT / 128) x 2 grid over the tile: 256 threads 2 x 2 (the default), 512 threads 4 x 2 --
  // the same tile and LDS with two waves per SIMD, for launches that are limited to one workgroup per CU by their LDS
  static_assert(NT == 256 || NT == 512, "gemm_body: 4 or 8 waves");
  constexpr int WROWS = NT / 128;
  constexpr int WM = BM / WROWS, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  static_assert(TM >= 1 && TN >= 1, "gemm_body: tile too small for the wave grid");
  using LA = LdsLayout<AKC, BM, BK>;
  using LB = LdsLayout<BKC, BN, BK>;
  // two LDS stages: slab s is consumed from stage s&1 while slab s+1 is written to the other one,
  // so one barrier per slab is enough and the staging writes overlap the MFMAs
  constexpr int kStage = ((LA::kSize + 3) & ~3) + ((LB::kSize + 3) & ~3);

  const int tiles_n = (p.N + BN - 1) / BN;
  int tm = __builtin_amdgcn_readfirstlane(tile_id / tiles_n), tn = __builtin_amdgcn_readfirstlane(tile_id % tiles_n);
  if (p.group_m > 1) {
    // grouped order: down group_m tile rows before moving one tile column on, so that the ~64 tiles an XCD works on at a
    // time form a compact 2-D block sharing its A and B panels in that XCD's L2 (in row-major order they are one tile row:
    // one A panel, 64 different B panels -- 1.5 GB fetched by the [20480 x 784] x [784 x 8192] K_uf tile for 90 MB of operands)
    const int tiles_m = (p.M + BM - 1) / BM;
    const int per_group = p.group_m * tiles_n;
    const int g = tile_id / per_group, first_m = g * p.group_m;
    const int gsize = min(tiles_m - first_m, p.group_m);
    const int rem = tile_id - g * per_group;
    tm = __builtin_amdgcn_readfirstlane(first_m + rem % gsize);
    tn = __builtin_amdgcn_readfirstlane(rem / gsize);
  }
  // triangular operands clip the K range per tile: hand out the long tiles of a matrix first, so that the launch does not
  // end on them (workgroups are dispatched in id order)
  if (p.triA == 1) tm = (p.M + BM - 1) / BM - 1 - tm;
  if (p.triB == 2) tn = tiles_n - 1 - tn;
  const int m0 = tm * BM, n0 = tn * BN;
  const int b = batch_id;
  const int i2 = __builtin_amdgcn_readfirstlane(b % p.nb2), i1 = __builtin_amdgcn_readfirstlane((b / p.nb2) % p.nb1),
            i0 = __builtin_amdgcn_readfirstlane(b / (p.nb2 * p.nb1));
  const float* A = p.A + i0 * p.sA[0] + i1 * p.sA[1] + i2 * p.sA[2];
  const float* B = p.B + i0 * p.sB[0] + i1 * p.sB[1] + i2 * p.sB[2];
  float* C = p.C + i0 * p.sC[0] + i1 * p.sC[1] + i2 * p.sC[2];
  const float* D = p.D ? p.D + i0 * p.sD[0] + i1 * p.sD[1] + i2 * p.sD[2] : nullptr;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;

  // tiles strictly above the diagonal of a lower-triangular result
  if (p.triC != 0 && n0 >= m0 + BM) {
    if (p.triC == 1) {
      for (int e = tid; e < BM * BN; e += NT) {
        const int r = m0 + e / BN, c = n0 + e % BN;
        if (r < p.M && c < p.N) C[(int64_t)r * p.ldc + c] = 0.f;
      }
    }
    return;
  }

  int ks = 0, ke = p.K;
  if (p.splitk > 1) {                        // this workgroup's BK-aligned share of K
    const int nslab = (p.K + BK - 1) / BK;
    const int per = (nslab + p.splitk - 1) / p.splitk;
    ks = min(p.K, split_id * per * BK);
    ke = min(p.K, (split_id + 1) * per * BK);
    C += (int64_t)split_id * p.sSplit;
  }
  if (p.triA == 1) ke = min(ke, m0 + BM);
  if (p.triA == 2) ks = max(ks, m0);
  if (p.triB == 1) ks = max(ks, n0);
  if (p.triB == 2) ke = min(ke, n0 + BN);
  ks = (ks / BK) * BK;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int c = 0; c < TN; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

  constexpr bool SC = RBF && SCALED;
  const float* kscale = SC ? p.kscale + i0 * p.ks_ld : nullptr;

  float ra[BM * BK / NT], rb[BN * BK / NT], rs[4] = {1.f, 1.f, 1.f, 1.f};
  float* const stage0 = lds;
  constexpr int kBoff = (LA::kSize + 3) & ~3;
  constexpr int NG = BK / 8;      // groups of 8 k: per group one float4 per lane and operand feeds 4 x TM*TN MFMAs

  // MFMAs of one slab held in LDS stage `As/Bs`.  Fragment reads run PF groups ahead of their MFMAs (one wave per
  // SIMD: nothing else hides the LDS latency); `between(g)` is called after the MFMAs of group g so the caller
  // can place staging work there.
  constexpr int PF = (NG >= 4) ? 2 : 1;
  auto slab_mfma = [&](const float* As, const float* Bs, auto&& between) {
    float af[NG][TM][4], bf[NG][TN][4];
    auto frag = [&](int g) {
      const int k = 8 * g + 4 * lh;
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        if constexpr (AKC) {
          const float4 v = *reinterpret_cast<const float4*>(&As[LA::at(wm0 + 32 * a + li, k)]);
          af[g][a][0] = v.x; af[g][a][1] = v.y; af[g][a][2] = v.z; af[g][a][3] = v.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) af[g][a][j] = As[LA::at(wm0 + 32 * a + li, k + j)];
        }
      }
#pragma unroll
      for (int c = 0; c < TN; ++c) {
        if constexpr (BKC) {
          const float4 v = *reinterpret_cast<const float4*>(&Bs[LB::at(wn0 + 32 * c + li, k)]);
          bf[g][c][0] = v.x; bf[g][c][1] = v.y; bf[g][c][2] = v.z; bf[g][c][3] = v.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) bf[g][c][j] = Bs[LB::at(wn0 + 32 * c + li, k + j)];
        }
      }
    };
#pragma unroll
    for (int g = 0; g < PF && g < NG; ++g) frag(g);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + PF < NG) frag(g + PF);
      if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);      // (see PIN above)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int c = 0; c < TN; ++c)
            acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g][a][j], bf[g][c][j], acc[a][c], 0, 0, 0);
      between(g);
    }
  };

  const int nfull = (ke > ks) ? (ke - ks) / BK : 0;      // slabs that need no k guard
  bool fastwg = false;                                      // uniform: bare float4 staging is legal for this workgroup
  const int64_t extA = AKC ? ((int64_t)(p.M - 1) * p.lda + p.K) : ((int64_t)(p.K - 1) * p.lda + p.M);
  const int64_t extB = BKC ? ((int64_t)(p.N - 1) * p.ldb + p.K) : ((int64_t)(p.K - 1) * p.ldb + p.N);
  if constexpr (VEC)
    fastwg = !p.nofast && nfull >= 1 && (AKC || (p.M % 4 == 0 && p.M >= 4)) && (BKC || (p.N % 4 == 0 && p.N >= 4)) &&
             extA < (1ll << 29) && extB < (1ll << 29);     // 32-bit byte offsets into each operand
  int kdone = ks;                                           // first k not yet accumulated

  if constexpr (VEC) {
    if (fastwg) {
      // ---- pipelined main loop over the full slabs: NO control flow around the loads/stores (a branch there
      // makes the compiler drain vmcnt before every load).  Loads past the last full slab re-read that slab and
      // the matching stores go to the idle LDS stage, where nobody reads them.
      constexpr int NPA = Pieces<AKC, BM, BK, true, NT>::kCount, NPB = Pieces<BKC, BN, BK, true, NT>::kCount, NP = NPA + NPB;
      constexpr int HALF = (NG >= 2) ? NG / 2 : 1;         // groups [0, HALF): global loads, [HALF, NG): LDS stores
      constexpr int PERL = (NP + HALF - 1) / HALF;          // pieces loaded per group
      constexpr int PERS = (NP + (NG - HALF > 0 ? NG - HALF : 1) - 1) / (NG - HALF > 0 ? NG - HALF : 1);
      int offA[NPA], offB[NPB];
#pragma unroll
      for (int c = 0; c < NPA; ++c) offA[c] = piece_offset<AKC, BM, BK, NT>(p.lda, m0, p.M, c);
#pragma unroll
      for (int c = 0; c < NPB; ++c) offB[c] = piece_offset<BKC, BN, BK, NT>(p.ldb, n0, p.N, c);
      const int stepA = 4 * (AKC ? BK : BK * p.lda), stepB = 4 * (BKC ? BK : BK * p.ldb);     // bytes per slab
      const i32x4 rsA = make_rsrc(A + (AKC ? ks : (int64_t)ks * p.lda), (int)(4 * extA));
      const i32x4 rsB = make_rsrc(B + (BKC ? ks : (int64_t)ks * p.ldb), (int)(4 * extB));
#pragma unroll
      for (int c = 0; c < NPA; ++c) offA[c] *= 4;
#pragma unroll
      for (int c = 0; c < NPB; ++c) offB[c] *= 4;
      // Two register sets: while slab sl is multiplied out of LDS, slab sl+1 goes registers(set X) -> idle LDS
      // stage and slab sl+2 goes global -> registers(set Y).  A load therefore has more than a whole slab of
      // MFMAs before its data is needed (one wave per SIMD cannot hide a global-load latency any other way).
      float ra2[BM * BK / NT], rb2[BN * BK / NT], rs2[4] = {1.f, 1.f, 1.f, 1.f};
      auto load_set = [&](int slab, float (&xa)[BM * BK / NT], float (&xb)[BN * BK / NT], float (&xs)[4]) {
#pragma unroll
        for (int c = 0; c < NPA; ++c) load_piece_fast<BM, BK, NT>(rsA, offA[c], slab * stepA, c, xa);
#pragma unroll
        for (int c = 0; c < NPB; ++c) load_piece_fast<BN, BK, NT>(rsB, offB[c], slab * stepB, c, xb);
        if constexpr (SC) load_scale<BK, true>(kscale, ks + slab * BK, ke, xs);
      };
      load_set(0, ra, rb, rs);
      store_slab<AKC, BM, BK, true, SC, NT>(stage0, ra, rs);
      store_slab<BKC, BN, BK, true, false, NT>(stage0 + kBoff, rb, rs);
      load_set(min(1, nfull - 1), ra, rb, rs);              // slab 1 -> set X
      __syncthreads();
      // The loop is unrolled by two, so the LDS stage each half works on is a compile-time constant and every LDS
      // address is a loop-invariant register plus an immediate.
      auto iteration = [&](auto stage_c, int sl, float (&xa)[BM * BK / NT], float (&xb)[BN * BK / NT], float (&xs)[4],
                           float (&ya)[BM * BK / NT], float (&yb)[BN * BK / NT], float (&ys)[4]) {
        constexpr int stage = decltype(stage_c)::value;
        const float* As = lds + stage * kStage;
        float* An = lds + (stage ^ 1) * kStage;
        const int nxt = min(sl + 2, nfull - 1);
        const int soA = nxt * stepA, soB = nxt * stepB;
        slab_mfma(As, As + kBoff, [&](int g) {
          if (g < HALF) {                                   // global (slab sl+2) -> register set Y
#pragma unroll
            for (int u = 0; u < PERL; ++u) {
              const int pc = g * PERL + u;
              if (pc < NPA) load_piece_fast<BM, BK, NT>(rsA, offA[pc], soA, pc, ya);
              else if (pc < NP) load_piece_fast<BN, BK, NT>(rsB, offB[pc - NPA], soB, pc - NPA, yb);
            }
            if constexpr (SC) { if (g == HALF - 1) load_scale<BK, true>(kscale, ks + nxt * BK, ke, ys); }
          }
          if (g >= HALF) {                                  // register set X (slab sl+1) -> idle LDS stage
#pragma unroll
            for (int u = 0; u < PERS; ++u) {
              const int pc = (g - HALF) * PERS + u;
              if (pc < NPA) store_piece<AKC, BM, BK, true, SC, NT>(An, xa, xs, pc);
              else if (pc < NP) store_piece<BKC, BN, BK, true, false, NT>(An + kBoff, xb, xs, pc - NPA);
            }
          }
        });
        __syncthreads();
      };
      // Two things keep the compiler's wait-count pass from putting `s_waitcnt vmcnt` at the loop HEADER, where it would wait,
      // every pair of slabs, for loads issued a few hundred cycles earlier (found in the ISA: `s_waitcnt vmcnt(2)` / `vmcnt(1)`
      // in front of the first two fragment reads -- one exposed L2 round trip per two slabs): (i) the fragment registers of the
      // first half alias register set X, which is dead there in the steady state but still pending on any path that reaches the
      // header with set X's loads in flight -- the loop entry (slab 1's loads) and, statically, an exit test between the two
      // halves that leaves through the latch.  So: slab 1's loads are waited for BEFORE the loop (once per tile), and the loop
      // body is always both halves; an odd last slab runs behind the loop.
      __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0) only (expcnt / lgkmcnt fields: no wait)
      int sl = 0;
      for (; sl + 1 < nfull; sl += 2) {
        iteration(std::integral_constant<int, 0>{}, sl, ra, rb, rs, ra2, rb2, rs2);
        iteration(std::integral_constant<int, 1>{}, sl + 1, ra2, rb2, rs2, ra, rb, rs);
      }
      if (sl < nfull) iteration(std::integral_constant<int, 0>{}, sl, ra, rb, rs, ra2, rb2, rs2);
      kdone = ks + nfull * BK;
    }
  }
  // ---- generic guarded slabs: everything when the fast path does not apply, otherwise only the K tail
  for (int k0 = kdone; k0 < ke; k0 += BK) {
    load_slab<AKC, BM, BK, VEC, NT>(A, p.lda, m0, p.M, k0, ke, ra);
    load_slab<BKC, BN, BK, VEC, NT>(B, p.ldb, n0, p.N, k0, ke, rb);
    if constexpr (SC) load_scale<BK, VEC>(kscale, k0, ke, rs);
    __syncthreads();                                        // previous slab fully consumed
    store_slab<AKC, BM, BK, VEC, SC, NT>(stage0, ra, rs);
    store_slab<BKC, BN, BK, VEC, false, NT>(stage0 + kBoff, rb, rs);
    __syncthreads();
    slab_mfma(stage0, stage0 + kBoff, [](int) {});
  }

  // epilogue
  float g2 = 0.f;
  const float *na = nullptr, *nbv = nullptr;
  if constexpr (RBF) {
    g2 = p.g2[i0];
    na = p.na + i0 * p.sNa[0] + i1 * p.sNa[1] + i2 * p.sNa[2];
    nbv = p.nbv + i0 * p.sNb[0] + i1 * p.sNb[1] + i2 * p.sNb[2];
  }
  if (p.symout && p.splitk <= 1) {
    // Symmetric result from the tiles that touch its lower triangle: C_ij = C_ji = value for j <= i.  The direct store is
    // coalesced as it stands (lanes run along a row); the mirrored one would scatter 4-byte writes down a column, so each
    // 32 x 32 accumulator block is transposed through LDS first (a private 32 x 33 patch per wave: conflict-free both
    // ways) and then stored along rows as well.  (Uniform control flow: every wave of the workgroup runs all blocks.)
    float* patch = lds + wave * (32 * 33);
    float dadd = 0.f;
    if constexpr (!RBF) { if (p.diag_ptr) dadd = p.diag_scale * p.diag_ptr[0]; }
    __syncthreads();                                          // the K loop is done with the LDS stages
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
      for (int c = 0; c < TN; ++c) {
        const int R0 = m0 + wm0 + 32 * a, C0 = n0 + wn0 + 32 * c;
        const int col = C0 + li;
        // norms up front on clamped indices (a load inside a bounds branch is a memory round trip of its own: 17 of them in
        // a row per block made this epilogue as long as the K loop); rows / columns past the edge are never stored
        float nbc = 0.f, nar[16];
        if constexpr (RBF) {
          nbc = nbv[min(col, p.N - 1)];
#pragma unroll
          for (int r = 0; r < 16; ++r) nar[r] = na[min(R0 + (r & 3) + 8 * (r >> 2) + 4 * lh, p.M - 1)];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rl = (r & 3) + 8 * (r >> 2) + 4 * lh, row = R0 + rl;
          float v;
          if constexpr (RBF) {
            const float d2 = nar[r] + nbc - 2.f * acc[a][c][r];
            v = (p.same_xy && row == col) ? g2 : g2 * expf(-0.5f * d2);
          } else {
            v = p.alpha * acc[a][c][r];
            if (D && row < p.M && col < p.N) v += p.beta * D[(int64_t)row * p.ldd + col];
            if (row == col && row >= p.diag_from) v += dadd;
          }
          if (row < p.M && col < p.N && col <= row) C[(int64_t)row * p.ldc + col] = v;
          patch[rl * 33 + li] = v;
        }
        __syncthreads();
        if (C0 < R0 + 32) {                                   // (uniform) the block has entries below the diagonal
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int jl = (r & 3) + 8 * (r >> 2) + 4 * lh;   // mirrored entry: row C0 + jl, column R0 + li
            const int mrow = C0 + jl, mcol = R0 + li;
            if (mrow < mcol && mcol < p.M && mrow < p.N) C[(int64_t)mrow * p.ldc + mcol] = patch[li * 33 + jl];
          }
        }
        __syncthreads();
      }
    }
    return;
  }
  // Interior tiles (the whole BM x BN tile inside the result: a workgroup-uniform test) take a straight-line epilogue: every
  // load up front, then arithmetic, then unconditional stores.  With a per-element bounds check each element becomes a
  // branch, and the compiler puts an `s_waitcnt vmcnt(0)` into every one of them -- which also waits for the PREVIOUS
  // element's store (measured on the stress K_uf tile: 64 exposed store round trips per thread, 30 % of the kernel; the
  // same for every accumulating product, whose D loads sat behind the branch).
  const bool full = (m0 + BM <= p.M) && (n0 + BN <= p.N);
  if constexpr (RBF) {
    if (p.splitk <= 1 && full) {
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        float nar[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) nar[r] = na[m0 + wm0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * lh];
#pragma unroll
        for (int c = 0; c < TN; ++c) {
          const int col = n0 + wn0 + 32 * c + li;
          const float nbc = nbv[col];
          float v[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float d2 = nar[r] + nbc - 2.f * acc[a][c][r];
            v[r] = (p.same_xy && row == col) ? g2 : g2 * expf(-0.5f * d2);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * lh;
            C[(int64_t)row * p.ldc + col] = v[r];
          }
        }
      }
      return;
    }
    if (p.splitk <= 1) {
      // edge tiles: row norms loaded up front with clamped indices, masked stores
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        float nar[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) nar[r] = na[min(m0 + wm0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * lh, p.M - 1)];
#pragma unroll
        for (int c = 0; c < TN; ++c) {
          const int col = n0 + wn0 + 32 * c + li;
          const float nbc = nbv[min(col, p.N - 1)];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float d2 = nar[r] + nbc - 2.f * acc[a][c][r];
            const float v = (p.same_xy && row == col) ? g2 : g2 * expf(-0.5f * d2);
            float* dst = (row < p.M && col < p.N) ? &C[(int64_t)row * p.ldc + col] : &g_gemm_trash[tid];
            *dst = v;
          }
        }
      }
      return;
    }
  }
  if constexpr (!RBF) {
    if (full && p.splitk <= 1) {      // plain product, interior tile (symout was handled above)
#pragma unroll
      for (int a = 0; a < TM; ++a) {
#pragma unroll
        for (int c = 0; c < TN; ++c) {
          const int col = n0 + wn0 + 32 * c + li;
          // (four rows at a time: the D values of a group are in flight together, and the kernel stays under 256 VGPRs,
          //  i.e. two workgroups per CU)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int rbase = m0 + wm0 + 32 * a + 8 * q + 4 * lh;
            float dv[4] = {0.f, 0.f, 0.f, 0.f};
            if (D) {
#pragma unroll
              for (int j = 0; j < 4; ++j) dv[j] = D[(int64_t)(rbase + j) * p.ldd + col];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float v = p.alpha * acc[a][c][4 * q + j];
              if (D) v += p.beta * dv[j];
              if (p.triC == 1 && col > rbase + j) v = 0.f;
              C[(int64_t)(rbase + j) * p.ldc + col] = v;
            }
          }
        }
      }
      return;
    }
  }
  if constexpr (!RBF) {
    if (p.splitk <= 1) {      // plain product, edge tile: straight-line, clamped D loads, out-of-range lanes store to the dump
#pragma unroll
      for (int a = 0; a < TM; ++a) {
#pragma unroll
        for (int c = 0; c < TN; ++c) {
          const int col = n0 + wn0 + 32 * c + li;
          const int colc = min(col, p.N - 1);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int rbase = m0 + wm0 + 32 * a + 8 * q + 4 * lh;
            float dv[4] = {0.f, 0.f, 0.f, 0.f};
            if (D) {
#pragma unroll
              for (int j = 0; j < 4; ++j) dv[j] = D[(int64_t)min(rbase + j, p.M - 1) * p.ldd + colc];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int row = rbase + j;
              float v = p.alpha * acc[a][c][4 * q + j];
              if (D) v += p.beta * dv[j];
              if (p.triC == 1 && col > row) v = 0.f;
              float* dst = (row < p.M && col < p.N) ? &C[(int64_t)row * p.ldc + col] : &g_gemm_trash[tid];
              *dst = v;
            }
          }
        }
      }
      return;
    }
  }
#pragma unroll
  for (int a = 0; a < TM; ++a) {
#pragma unroll
    for (int c = 0; c < TN; ++c) {
      const int col = n0 + wn0 + 32 * c + li;
      float nbc = 0.f;
      if constexpr (RBF) nbc = (col < p.N) ? nbv[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < p.M && col < p.N) {
          float v;
          if constexpr (RBF) {
            if (p.splitk > 1) {
              v = acc[a][c][r];               // partial inner product; rbf_combine_kernel finishes the job
            } else {
              const float d2 = na[row] + nbc - 2.f * acc[a][c][r];
              v = (p.same_xy && row == col) ? g2 : g2 * expf(-0.5f * d2);
            }
          } else {
            v = p.alpha * acc[a][c][r];
            if (D) v += p.beta * D[(int64_t)row * p.ldd + col];
            if (p.triC == 1 && col > row) v = 0.f;
            if (p.splitk > 1) {               // K split without a scratch buffer: partial sums meet in C (pre-zeroed)
              if (v != 0.f) atomicAdd(&C[(int64_t)row * p.ldc + col], v);
              continue;
            }
          }
          C[(int64_t)row * p.ldc + col] = v;
        }
      }
    }
  }
}

// Persistent variant of gemm_body's fast path for PLAIN products (no RBF epilogue, no D, no split-K, no triangular hints; K a
// multiple of BK with at least four slabs; operands as the fast path wants them -- gemm_persist_ok() on the host).  A workgroup
// takes tile after tile from a work queue and treats the K slabs of ALL its tiles as one pipelined sequence: while the last
// slabs of a tile are multiplied the first slabs of the next one are already on their way (global -> registers -> idle LDS
// stage), and a tile's result is stored between two slabs.  Why: a 64 x 64 x 512 tile of the backward's P_uf product takes
// 13 us as a workgroup of its own against 6.8 us of MFMAs -- descriptor set-up, the cold round trip of the first two slabs and
// the epilogue are paid per tile, three times per CU.
// Queue: `queue` = 8 counters (zero when the kernel starts), one per XCD (x = blockIdx.x % 8: workgroups go to the XCDs
// round-robin); XCD x hands out the contiguous range [x T / 8, (x + 1) T / 8) of the (batch, tile_m, tile_n) order, so the
// tiles in flight on one L2 share their A panels (as xcd_remap does for one-tile workgroups).  Any workgroup of the launch
// may join at any time -- the matrix-chain workgroups of t0_bwdmat_gemm_kernel do when their chain is finished (with many
// hyper-samples the chains end long before the product).  The next tile's id is fetched (thread 0, one atomic) during the first
// slab of the current tile and published through LDS a few slabs later (iteration()).
// rank >= 0: this workgroup's first tile is number `rank` of its XCD's range (no round trip to the queue before the first
// load), the queue hands out the tiles from number `nstatic` on (= the number of workgroups that start this way on the XCD);
// rank < 0: a late joiner, which asks the queue for its first tile too.
// queue == NULL: no queue -- the workgroup's tiles are rank, rank + nstatic, rank + 2 nstatic, ... of the range.  (The
// returning atomic costs: the loads issued behind it cannot retire before it does -- vmcnt counts in order -- and a
// device-scope atomic takes longer than a slab.  P_uf alone at S = 3: 39.4 us static, 44.0 with the queue; with 8+ samples,
// where the chains' CUs would otherwise idle for most of the launch, the queue wins: S = 8 step 496 -> 476 us.)
template <int BM, int BN, int BK, bool AKC, bool BKC>
__device__ __forceinline__ void gemm_persist_body(const GemmParams& p, int* __restrict__ queue, const int tiles, const int total,
                                                  float* __restrict__ lds, const int rank, const int nstatic) {
  constexpr int NT = 256;
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  using LA = LdsLayout<AKC, BM, BK>;
  using LB = LdsLayout<BKC, BN, BK>;
  constexpr int kBoff = (LA::kSize + 3) & ~3;
  constexpr int kStage = kBoff + ((LB::kSize + 3) & ~3);
  constexpr int NG = BK / 8, PF = (NG >= 4) ? 2 : 1;
  const int x = (int)blockIdx.x & 7;
  const int lo = (int)((int64_t)total * x / 8), hi = (int)((int64_t)total * (x + 1) / 8);
  int* const qx = queue + x;
  int* const slot = reinterpret_cast<int*>(lds + 2 * kStage);          // one word behind the two stages
  const int tiles_n = (p.N + BN - 1) / BN;
  const int ns = p.K / BK;                                 // slabs per tile (>= 4)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;

  const int q0 = lo + nstatic;                             // the queue's first tile
  int idM;                                                 // tile being multiplied
  if (rank >= 0) {
    idM = lo + rank < hi ? lo + rank : -1;
  } else {                                                 // fetched and published before anything else
    __syncthreads();                                       // (a caller that used the LDS before is done with it)
    if (tid == 0) { const int t = q0 + atomicAdd(qx, 1); *slot = t < hi ? t : -1; }
    __syncthreads();
    idM = __builtin_amdgcn_readfirstlane(*slot);
  }
  if (idM < 0) return;
  if (queue == nullptr && rank < 0) return;
  int idL = idM;                                           // tile whose slabs are being fetched
  int idN = -2;                                            // the tile after idM: -2 not known yet, -1 none

  constexpr int NPA = Pieces<AKC, BM, BK, true>::kCount, NPB = Pieces<BKC, BN, BK, true>::kCount, NP = NPA + NPB;
  constexpr int HALF = (NG >= 2) ? NG / 2 : 1;
  constexpr int PERL = (NP + HALF - 1) / HALF;
  constexpr int PERS = (NP + (NG - HALF > 0 ? NG - HALF : 1) - 1) / (NG - HALF > 0 ? NG - HALF : 1);
  const int64_t extA = AKC ? ((int64_t)(p.M - 1) * p.lda + p.K) : ((int64_t)(p.K - 1) * p.lda + p.M);
  const int64_t extB = BKC ? ((int64_t)(p.N - 1) * p.ldb + p.K) : ((int64_t)(p.K - 1) * p.ldb + p.N);
  const int stepA = 4 * (AKC ? BK : BK * p.lda), stepB = 4 * (BKC ? BK : BK * p.ldb);     // bytes per slab

  struct Tile { int m0, n0, i0, i1, i2; };
  auto locate = [&](int id) {
    const int b = id / tiles, t = id - b * tiles;
    Tile r;
    r.m0 = (t / tiles_n) * BM; r.n0 = (t % tiles_n) * BN;
    r.i2 = b % p.nb2; r.i1 = (b / p.nb2) % p.nb1; r.i0 = b / (p.nb2 * p.nb1);
    return r;
  };
  i32x4 rsA, rsB;
  int offA[NPA], offB[NPB];
  auto set_load_tile = [&](int id) {
    const Tile t = locate(id);
    rsA = make_rsrc(p.A + t.i0 * p.sA[0] + t.i1 * p.sA[1] + t.i2 * p.sA[2], (int)(4 * extA));
    rsB = make_rsrc(p.B + t.i0 * p.sB[0] + t.i1 * p.sB[1] + t.i2 * p.sB[2], (int)(4 * extB));
#pragma unroll
    for (int c = 0; c < NPA; ++c) offA[c] = 4 * piece_offset<AKC, BM, BK>(p.lda, t.m0, p.M, c);
#pragma unroll
    for (int c = 0; c < NPB; ++c) offB[c] = 4 * piece_offset<BKC, BN, BK>(p.ldb, t.n0, p.N, c);
  };
  int sL = 0;                                              // slab of the NEXT load (of tile idL)
  auto advance_load = [&]() {
    if (sL + 1 < ns) { ++sL; return; }
    // the tile is fetched completely (this is the iteration that multiplies its slab ns - 3): on to the next one, if the queue
    // had one (idN was published with the barrier of slab ns - 4); otherwise the last slab is fetched again (and never used)
    if (idL == idM && idN >= 0) { idL = idN; sL = 0; set_load_tile(idL); }
  };

  f32x16 acc[TM][TN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int c = 0; c < TN; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
  };
  zero_acc();

  auto slab_mfma = [&](const float* As, const float* Bs, auto&& between) {
    float af[NG][TM][4], bf[NG][TN][4];
    auto frag = [&](int g) {
      const int k = 8 * g + 4 * lh;
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        if constexpr (AKC) {
          const float4 v = *reinterpret_cast<const float4*>(&As[LA::at(wm0 + 32 * a + li, k)]);
          af[g][a][0] = v.x; af[g][a][1] = v.y; af[g][a][2] = v.z; af[g][a][3] = v.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) af[g][a][j] = As[LA::at(wm0 + 32 * a + li, k + j)];
        }
      }
#pragma unroll
      for (int c = 0; c < TN; ++c) {
        if constexpr (BKC) {
          const float4 v = *reinterpret_cast<const float4*>(&Bs[LB::at(wn0 + 32 * c + li, k)]);
          bf[g][c][0] = v.x; bf[g][c][1] = v.y; bf[g][c][2] = v.z; bf[g][c][3] = v.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) bf[g][c][j] = Bs[LB::at(wn0 + 32 * c + li, k + j)];
        }
      }
    };
#pragma unroll
    for (int g = 0; g < PF && g < NG; ++g) frag(g);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + PF < NG) frag(g + PF);
      __builtin_amdgcn_sched_barrier(0);      // fragment prefetch stays in front of the MFMAs (gemm_body: PIN)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int c = 0; c < TN; ++c)
            acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g][a][j], bf[g][c][j], acc[a][c], 0, 0, 0);
      between(g);
    }
  };

  float ra[BM * BK / NT], rb[BN * BK / NT], ra2[BM * BK / NT], rb2[BN * BK / NT];
  const float one[4] = {1.f, 1.f, 1.f, 1.f};
  auto load_set = [&](float (&xa)[BM * BK / NT], float (&xb)[BN * BK / NT]) {
    const int soA = sL * stepA, soB = sL * stepB;
#pragma unroll
    for (int c = 0; c < NPA; ++c) load_piece_fast<BM, BK>(rsA, offA[c], soA, c, xa);
#pragma unroll
    for (int c = 0; c < NPB; ++c) load_piece_fast<BN, BK>(rsB, offB[c], soB, c, xb);
  };
  set_load_tile(idL);
  load_set(ra, rb);
  advance_load();
  store_slab<AKC, BM, BK, true, false>(lds, ra, one);
  store_slab<BKC, BN, BK, true, false>(lds + kBoff, rb, one);
  load_set(ra, rb);                                        // slab 1 -> set X
  advance_load();
  __syncthreads();
  // (slab 1's loads are waited for here, once per workgroup: pending at the loop entry they make the compiler put `s_waitcnt
  //  vmcnt` in front of the first fragment reads of EVERY iteration -- the fragment registers alias set X -- see gemm_body)
  __builtin_amdgcn_s_waitcnt(0x0F70);

  int sM = 0;                                              // slab being multiplied (of tile idM)
  int fetched = 0;                                         // thread 0: the queue's answer, in flight
  // the result of the tile the multiply cursor is on: straight-line stores, lanes outside the matrix write to the dump
  auto epilogue = [&]() {
    const Tile t = locate(idM);
    float* C = p.C + t.i0 * p.sC[0] + t.i1 * p.sC[1] + t.i2 * p.sC[2];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
      for (int c = 0; c < TN; ++c) {
        const int col = t.n0 + wn0 + 32 * c + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = t.m0 + wm0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * lh;
          float* dst = (row < p.M && col < p.N) ? &C[(int64_t)row * p.ldc + col] : &g_gemm_trash[tid];
          *dst = p.alpha * acc[a][c][r];
        }
      }
    }
  };
  // one slab; returns false when the last tile of this workgroup is done
  auto iteration = [&](auto stage_c, float (&xa)[BM * BK / NT], float (&xb)[BN * BK / NT], float (&ya)[BM * BK / NT],
                       float (&yb)[BN * BK / NT]) -> bool {
    constexpr int stage = decltype(stage_c)::value;
    const float* As = lds + stage * kStage;
    float* An = lds + (stage ^ 1) * kStage;
    const int soA = sL * stepA, soB = sL * stepB;
    // (uniform) first slab of a tile: thread 0 asks the queue for the tile after it; the answer -- a device-scope atomic takes
    // longer than a slab -- is published through LDS with the barrier of slab ns - 4, one slab before the load cursor needs it
    const bool publish = sM == ns - 4;
    if (queue != nullptr && sM == 0 && tid == 0) fetched = atomicAdd(qx, 1);
    slab_mfma(As, As + kBoff, [&](int g) {
      if (g < HALF) {                                      // global (two slabs ahead) -> register set Y
#pragma unroll
        for (int u = 0; u < PERL; ++u) {
          const int pc = g * PERL + u;
          if (pc < NPA) load_piece_fast<BM, BK>(rsA, offA[pc], soA, pc, ya);
          else if (pc < NP) load_piece_fast<BN, BK>(rsB, offB[pc - NPA], soB, pc - NPA, yb);
        }
      }
      if (g >= HALF) {                                     // register set X (one slab ahead) -> idle LDS stage
#pragma unroll
        for (int u = 0; u < PERS; ++u) {
          const int pc = (g - HALF) * PERS + u;
          if (pc < NPA) store_piece<AKC, BM, BK, true, false>(An, xa, one, pc);
          else if (pc < NP) store_piece<BKC, BN, BK, true, false>(An + kBoff, xb, one, pc - NPA);
        }
      }
    });
    if (queue != nullptr && publish && tid == 0) { const int t = q0 + fetched; *slot = t < hi ? t : -1; }
    advance_load();
    bool more = true;
    if (++sM == ns) {                                      // (uniform) the tile is complete
      epilogue();
      zero_acc();
      sM = 0;
      more = idN >= 0;
      idM = idN; idN = -2;
    }
    __syncthreads();
    if (publish) {
      if (queue != nullptr) idN = __builtin_amdgcn_readfirstlane(*slot);
      else idN = idM + nstatic < hi ? idM + nstatic : -1;      // static list: every nstatic-th tile of the range
    }
    return more;
  };
  for (;;) {
    if (!iteration(std::integral_constant<int, 0>{}, ra, rb, ra2, rb2)) break;
    if (!iteration(std::integral_constant<int, 1>{}, ra2, rb2, ra, rb)) break;
  }
}

// Workgroup -> tile map that keeps each XCD on a compact set of tiles.  Workgroups go to the 8 XCDs round-robin by
// linear id, and each XCD has its own L2: with the plain map the 8 column-tiles that share an A panel land on 8
// different L2s and every panel is fetched 8 times (measured: 98 MB fetched by the backward pair GEMM against 12 MB of
// operands).  Here XCD x works through the contiguous range [x T/8, (x+1) T/8) of the (batch, tile_m, tile_n) order
// instead, so the tiles of one A panel meet in one L2.  A speed heuristic only: nothing depends on the placement.
__device__ __forceinline__ int xcd_remap(int lin, int total) {
  constexpr int kXcd = 8;
  const int q = total / kXcd, r = total % kXcd;
  const int xcd = lin % kXcd, idx = lin / kXcd;
  return xcd * q + (xcd < r ? xcd : r) + idx;
}

template <int BM, int BN, int BK, bool AKC, bool BKC, bool VEC, bool RBF, bool SCALED = true>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(16))) float lds[gemm_lds_floats<BM, BN, BK, AKC, BKC>()];
  STEP_SPAN(gemm, 2);
  const int lin = (int)(blockIdx.y * gridDim.x + blockIdx.x), total = (int)(gridDim.x * gridDim.y);
  const int id = p.xcd_remap ? xcd_remap(lin, total) : lin;
  gemm_body<BM, BN, BK, AKC, BKC, VEC, RBF, SCALED>(p, id % (int)gridDim.x, id / (int)gridDim.x, blockIdx.z, lds);
}

// the same with eight waves per workgroup (two per SIMD on the same tile: gemm_body's NT = 512), plain products only
template <int BM, int BN, int BK, bool AKC, bool BKC>
__global__ __launch_bounds__(512) void gemm_kernel_w8(const GemmParams p) {
  __shared__ __attribute__((aligned(16))) float lds[gemm_lds_floats<BM, BN, BK, AKC, BKC>()];
  const int lin = (int)(blockIdx.y * gridDim.x + blockIdx.x), total = (int)(gridDim.x * gridDim.y);
  const int id = p.xcd_remap ? xcd_remap(lin, total) : lin;
  gemm_body<BM, BN, BK, AKC, BKC, true, false, true, 512>(p, id % (int)gridDim.x, id / (int)gridDim.x, blockIdx.z, lds);
}

// Two independent problems of the same kernel flavour in ONE launch (1-D grid: the workgroups of problem 0, then
// those of problem 1).  Mid-size problems that cannot fill the chip alone (K_uu: 120 workgroups, K_uf: 384) share it.
struct GemmPair { GemmParams p[2]; int nwg0; int tiles[2]; int per_split[2]; };   // per_split = tiles * nbatch
template <int BM, int BN, int BK, bool AKC, bool BKC, bool VEC, bool RBF>
__global__ __launch_bounds__(256) void gemm_pair_kernel(const GemmPair pp) {
  // XCD-compact tile ranges within each problem (not across both: the two problems have different K, and a split
  // across both would hand whole XCDs only short or only long workgroups)
  const int which = (int)blockIdx.x >= pp.nwg0 ? 1 : 0;       // wave-uniform
  const int id = which ? xcd_remap((int)blockIdx.x - pp.nwg0, (int)gridDim.x - pp.nwg0) : xcd_remap((int)blockIdx.x, pp.nwg0);
  const int tiles = pp.tiles[which];
  __shared__ __attribute__((aligned(16))) float lds[gemm_lds_floats<BM, BN, BK, AKC, BKC>()];
  if (which == 0) gemm_body<BM, BN, BK, AKC, BKC, VEC, RBF>(pp.p[0], id % tiles, id / tiles, 0, lds);
  else gemm_body<BM, BN, BK, AKC, BKC, VEC, RBF>(pp.p[1], id % tiles, id / tiles, 0, lds);
}

// Same idea for two plain products with DIFFERENT operand layouts (e.g. gG = P gW^T next to gP += G gW): both layout
// instances live in one kernel, the workgroup picks by problem.
template <bool A0, bool B0, bool A1, bool B1>
__global__ __launch_bounds__(256) void gemm_pair2_kernel(const GemmPair pp) {
  const int which = (int)blockIdx.x >= pp.nwg0 ? 1 : 0;       // wave-uniform
  const int id = which ? xcd_remap((int)blockIdx.x - pp.nwg0, (int)gridDim.x - pp.nwg0) : xcd_remap((int)blockIdx.x, pp.nwg0);
  const int tiles = pp.tiles[which];
  __shared__ __attribute__((aligned(16))) float lds[cmax(gemm_lds_floats<64, 64, 64, A0, B0>(), gemm_lds_floats<64, 64, 64, A1, B1>())];
  const int split = id / pp.per_split[which], rem = id % pp.per_split[which];   // K-split outermost
  if (which == 0) gemm_body<64, 64, 64, A0, B0, true, false>(pp.p[0], rem % tiles, rem / tiles, split, lds);
  else gemm_body<64, 64, 64, A1, B1, true, false>(pp.p[1], rem % tiles, rem / tiles, split, lds);
}

template <int BM, int BN, int BK, bool VEC, bool RBF>
static void dispatch_layout(const GemmParams& p, int transA, int transB, dim3 grid, hipStream_t st) {
  // op(A) K-contiguous <=> transA == 0;  op(B) K-contiguous <=> transB == 1
  const bool akc = transA == 0, bkc = transB == 1;
  if constexpr (RBF) {
    if (p.kscale) hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, true, true, VEC, true, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, true, true, VEC, true, false>), grid, dim3(256), 0, st, p);
  } else {
    if constexpr (VEC && BM == 128 && BN == 64 && BK == 32) {
      // eight waves per 128 x 64 x 32 tile (two per SIMD; VARGP_GEMM_W8=0: four): Permuted-MNIST t = 1 / 4 / 9 step 3.58 -> 3.52 /
      // 13.82 -> 13.74 / 49.99 -> 49.72 ms, Split-MNIST t = 1 610 -> 608 us
      static const int w8 = [] { const char* e = getenv("VARGP_GEMM_W8"); return e ? atoi(e) : 1; }();   // tuning aid
      if (w8) {
        if (akc && bkc) hipLaunchKernelGGL((gemm_kernel_w8<BM, BN, BK, true, true>), grid, dim3(512), 0, st, p);
        else if (akc && !bkc) hipLaunchKernelGGL((gemm_kernel_w8<BM, BN, BK, true, false>), grid, dim3(512), 0, st, p);
        else if (!akc && bkc) hipLaunchKernelGGL((gemm_kernel_w8<BM, BN, BK, false, true>), grid, dim3(512), 0, st, p);
        else hipLaunchKernelGGL((gemm_kernel_w8<BM, BN, BK, false, false>), grid, dim3(512), 0, st, p);
        return;
      }
    }
    if (akc && bkc) hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, true, true, VEC, false>), grid, dim3(256), 0, st, p);
    else if (akc && !bkc) hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, true, false, VEC, false>), grid, dim3(256), 0, st, p);
    else if (!akc && bkc) hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, false, true, VEC, false>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, false, false, VEC, false>), grid, dim3(256), 0, st, p);
  }
}

template <int BM, int BN, int BK>
static void dispatch_tile(const GemmParams& p, int transA, int transB, int nbatch, bool rbf, bool vec, hipStream_t st) {
  dim3 grid(cdiv(p.M, BM) * cdiv(p.N, BN), nbatch, p.splitk > 1 ? p.splitk : 1);
  if (rbf) {
    if (vec) dispatch_layout<BM, BN, BK, true, true>(p, 0, 1, grid, st);
    else dispatch_layout<BM, BN, BK, false, true>(p, 0, 1, grid, st);
  } else {
    if (vec) dispatch_layout<BM, BN, BK, true, false>(p, transA, transB, grid, st);
    else dispatch_layout<BM, BN, BK, false, false>(p, transA, transB, grid, st);
  }
}


// Tile choice (MI355X: 256 CUs).  The f32 MFMA runs at the vector rate, so what matters is (a) filling
// the CUs and (b) amortising the per-slab staging + 2 barriers over enough MFMAs:
//   128x128xBK16 : 32 MFMA / slab / wave  - when that still gives >= ~1 workgroup per CU
//   128x64 xBK32 : 32 MFMA / slab / wave  - mid-size problems (K_uf at Split-MNIST: 192 workgroups)
//   64 x64 xBK64 :  32 MFMA / slab / wave - small problems; the K extent of the (M x M) products of the
//                  ELBO (K <= 128) is covered by one or two slabs, i.e. one or two global-load latencies.
// Split-K for mid-size RBF products is implemented (partials + rbf_combine_kernel) but OFF by default: at the
// Split-MNIST K_uf shape (384 tiles of 64x64 on 256 CUs) two splits measured 69.8 us against 61.6 us unsplit.
// The per-slab cost of a wave is its 32 MFMAs plus ~1000 cycles of VALU/LDS issue that do not overlap them, and
// waves sharing a SIMD serialise, so more, shorter workgroups only add prologue/epilogue and the combine pass.
// VARGP_RBF_SPLITK=2 turns it on for experiments.
int rbf_splitk(int M, int N, int K, int nbatch) {
  (void)M; (void)N; (void)K; (void)nbatch;
  static const int force = [] { const char* e = getenv("VARGP_RBF_SPLITK"); return e ? atoi(e) : 0; }();
  return force >= 2 ? 2 : 1;      // the workspace holds at most two partials
}

// One launch, two independent roles: workgroups [0, nchol) factorise one small matrix each (chol3_body: the K_uu + eps I
// and S_u + eps I factorisations of the ELBO, a latency-bound chain of n pivots on nchol CUs), the others are tiles of an
// RBF kernel-matrix GEMM (K_uf) that does not depend on the factorisations.  Both need 256 threads.  Launched
// separately the GEMM would wait for the factorisation kernel, which leaves five sixths of the chip idle.
struct CholArgs {
  const float* A; int lda; int64_t sA; float eps;
  float* L; int ldl; int64_t sL;
  float* T; int ldt; int64_t sT;
  int32_t* info; int n; int nchol;
  CholExtra extra;   // base == nullptr: none
  ZeroJobs zero;     // third role (the last nzero workgroups): zero-fills of the caller, free under the pivot chains
  int nzero;
};
// NT = 512 (round 4): the launch is held to one workgroup per CU by its LDS reservation, so the GEMM role runs one wave per
// SIMD at 256 threads; with 512 threads the same 128 x 64 tile is worked by 8 waves (two per SIMD); the factorisation and
// zero-fill roles use the first four waves (the others leave at once: a finished wave no longer counts at s_barrier).
template <int KC, int SETS, int BM, int BK, bool SCALED = true, class R = double, int NT = 256>
__global__ __launch_bounds__(NT) void chol_rbf_gemm_kernel(const CholArgs c, const GemmParams p, const int tiles) {
  // one LDS array for both roles (the factorisation stages its matrix through 40 KB of it): 2 workgroups per CU
  __shared__ __attribute__((aligned(16))) float lds[cmax(gemm_lds_floats<BM, 64, BK, true, true>(), chol3_stage_floats<KC>())];
  STEP_SPAN(gemm, 1);
  if ((int)blockIdx.x < c.nchol) {
    if (NT > 256 && threadIdx.x >= 256) return;
    chol3_body<KC, SETS, R>(blockIdx.x, c.A, c.lda, c.sA, c.eps, c.L, c.ldl, c.sL, c.T, c.ldt, c.sT, nullptr, c.info, 0, c.n,
                            0, lds, c.extra.base ? &c.extra : nullptr);
    STEP_SPAN_MARK(gemm, 9);
    return;
  }
  const int ngemm = (int)gridDim.x - c.nchol - c.nzero;
  if ((int)blockIdx.x >= c.nchol + ngemm) {
    if (NT > 256 && threadIdx.x >= 256) return;
    zero_jobs_role(c.zero, (int)blockIdx.x - c.nchol - ngemm, c.nzero);
    return;
  }
  const int id = xcd_remap((int)blockIdx.x - c.nchol, ngemm);
#ifdef VARGP_CHOL_PHASES
  if (threadIdx.x == 0 && ((int)blockIdx.x == c.nchol || (int)blockIdx.x == c.nchol + ngemm - 1))
    g_chol_phase[((int)blockIdx.x == c.nchol ? 32 : 40)] = __builtin_amdgcn_s_memrealtime();
#endif
  gemm_body<BM, 64, BK, true, true, true, true, SCALED, NT, true>(p, id % tiles, id / tiles, 0, lds);
#ifdef VARGP_CHOL_PHASES
  if (threadIdx.x == 0 && ((int)blockIdx.x == c.nchol || (int)blockIdx.x == c.nchol + ngemm - 1))
    g_chol_phase[((int)blockIdx.x == c.nchol ? 32 : 40) + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}
#ifdef VARGP_CHOL_PHASES
extern "C" void vargp_debug_chol_phases(unsigned long long* out, int last) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chol_phase), 64 * 8);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_chol_phase_last), &last, sizeof(int));
}
#endif

// Stand-alone blocked factorisation (chol.hip): the pivot chains of one diagonal block (one workgroup per matrix: TEN of the 256
// CUs at BASELINE config 5) next to a plain NN product that does not depend on them -- block row K of T = L^-1 to the LEFT of
// the outer block, B1 = L[K, 0:K0] T[0:K0, 0:K0], which only needs earlier panels.  fp64 chains (R) with the caller's
// info_base, exactly as chol_inv_small3_kernel; 64 x 64 x 64 tiles as gemm_kernel.
template <int KC, int SETS, class R>
__global__ __launch_bounds__(256) void chol_nn_gemm_kernel(const CholArgs c, const GemmParams p, const int tiles, const int info_base) {
  __shared__ __attribute__((aligned(16))) float lds[cmax(gemm_lds_floats<64, 64, 64, true, false>(), chol3_stage_floats<KC>())];
  if ((int)blockIdx.x < c.nchol) {
    chol3_body<KC, SETS, R>(blockIdx.x, c.A, c.lda, c.sA, c.eps, c.L, c.ldl, c.sL, c.T, c.ldt, c.sT, nullptr, c.info, info_base, c.n,
                            0, lds, nullptr);
    return;
  }
  const int ngemm = (int)gridDim.x - c.nchol - c.nzero;
  if ((int)blockIdx.x >= c.nchol + ngemm) {       // third role: zero-fills of the caller (the blocked driver's L / T above the block diagonal)
    zero_jobs_role(c.zero, (int)blockIdx.x - c.nchol - ngemm, c.nzero);
    return;
  }
  const int id = xcd_remap((int)blockIdx.x - c.nchol, ngemm);
  gemm_body<64, 64, 64, true, false, true, false>(p, id % tiles, id / tiles, 0, lds);
}

// One launch, two independent roles (like chol_rbf_gemm_kernel in the forward): workgroups [0, nmat) walk the chain of
// M x M products of one matrix of the first-task backward (t0_bwd_mat.h: ~20 us on nmat CUs), the others are 64 x 64 tiles of a
// plain NN product the chain does not feed (P_uf = W_uf x next to the K_uu matrices, P_uu = W_uu z next to the S_u ones).
// Every workgroup is carved the chain's 136 KB of LDS, i.e. one workgroup per CU.
__global__ __launch_bounds__(256) void t0_bwdmat_gemm_kernel(const BwdMatArgs a, const int first, const int nmat,
                                                             const GemmParams p, const int tiles, int* __restrict__ queue,
                                                             const int total, const int persist) {
  extern __shared__ __attribute__((aligned(16))) float bmat_lds[];
  STEP_SPAN(gemm, 5);
#ifdef STEP_SPANS
  const unsigned long long t_in_ = wall_clock64();
#endif
#ifdef BMAT_STAMPS      // tuning builds: wall-clock span (100 MHz) of each role over all its workgroups, [role][first start, last end, last start]
  const unsigned long long t_in = wall_clock64();
  const int role = (int)blockIdx.x < nmat ? 0 : 1;
  if (threadIdx.x == 0) { atomicMin(&g_bmat_span[role][0], t_in); atomicMax(&g_bmat_span[role][2], t_in); }
#endif
  if ((int)blockIdx.x < nmat) {
    t0_bwd_mat_body(a, first + (int)blockIdx.x, bmat_lds);
    STEP_SPAN_MARK(gemm, 8);
#ifdef STEP_SPANS
    if (threadIdx.x == 0 && blockIdx.x < 64) { g_bmat_ends[blockIdx.x][1] = wall_clock64(); g_bmat_ends[blockIdx.x][0] = t_in_; }
#endif
    // queue != NULL: the product's tiles come from a work queue (gemm_persist_body) -- a finished chain joins in
    // (workgroups go to the XCDs round-robin by blockIdx: XCD x holds cx of the chains and nx of the product's workgroups)
    const int x = (int)blockIdx.x & 7;
    const int cx = nmat > x ? (nmat - x + 7) >> 3 : 0, nx = (((int)gridDim.x - x + 7) >> 3) - cx;
    if (queue) gemm_persist_body<64, 64, 64, true, false>(p, queue, tiles, total, bmat_lds, -1, nx);
  } else if (persist) {
    const int x = (int)blockIdx.x & 7;
    const int cx = nmat > x ? (nmat - x + 7) >> 3 : 0, nx = (((int)gridDim.x - x + 7) >> 3) - cx;
    gemm_persist_body<64, 64, 64, true, false>(p, queue, tiles, total, bmat_lds, ((int)blockIdx.x >> 3) - cx, nx);
  } else {
    const int id = xcd_remap((int)blockIdx.x - nmat, (int)gridDim.x - nmat);
    gemm_body<64, 64, 64, true, false, true, false>(p, id % tiles, id / tiles, 0, bmat_lds);
  }
#ifdef BMAT_STAMPS
  if (threadIdx.x == 0) atomicMax(&g_bmat_span[role][1], wall_clock64());
#endif
}

static bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
static bool gemm_vec_ok(const GemmParams& p) {
  bool vec = aligned16(p.A) && aligned16(p.B) && (p.lda % 4 == 0) && (p.ldb % 4 == 0);
  for (int i = 0; i < 3; ++i) vec = vec && (p.sA[i] % 4 == 0) && (p.sB[i] % 4 == 0);
  return vec;
}

// One launch, three independent roles at the front of the first-task forward: the K-split inner products of K_uu (64 x 64 x BK
// tiles: the long pole, dispatched first), the prologue roles (t0_prologue.h) and the row norms of [z; x].  The product and
// the norms need 1/sigma^2 of their hyper-sample, which the prologue's theta role only writes in this same launch: every
// workgroup evaluates the entries it needs itself (same expression, so the copies in `w` agree bit for bit).
template <int BK>
__global__ __launch_bounds__(256) void t0_pro_kuu_kernel(const ProArgs a, const int npro, const NormArgs nr, const int nnorm,
                                                         const GemmParams p, const int tiles, const int ngemm,
                                                         const int norms_first) {
  __shared__ __attribute__((aligned(16))) float lds[gemm_lds_floats<64, 64, BK, true, true>()];
  STEP_SPAN(gemm, 0);
  int blk = blockIdx.x;
  if (blk >= ngemm) {      // (the other order -- short roles first -- measured 26.9 us against 20.3 us)
    blk -= ngemm;
    // norms_first: the norm role (the longer of the two: 16 rows x S per workgroup, x o w written) in front of the prologue roles
    const int pblk = norms_first ? blk - nnorm : blk, nblk = norms_first ? blk : blk - npro;
    if (pblk >= 0 && pblk < npro) t0_prologue_body(a, pblk, lds);
    else {
      if (step_span_guard_.p && threadIdx.x == 0) atomicMax(&g_spans_gemm[11][0], wall_clock64());      // (slot 11: LAST start .. last end)
      t0_norm_body(a, nr, nblk, lds); STEP_SPAN_MARK(gemm, 11);
    }
    return;
  }
  const int per = ngemm / p.splitk;            // tiles * nbatch
  const int split = blk / per, r = blk - split * per;
  const int batch = r / tiles, tile = r - batch * tiles;
  {  // this workgroup's K range (gemm_body's rule) of w[s, :], s = first batch index
    const int s = batch / (p.nb2 * p.nb1);
    const int nslab = (p.K + BK - 1) / BK, share = (nslab + p.splitk - 1) / p.splitk;
    const int ks = min(p.K, split * share * BK), ke = min(p.K, (split + 1) * share * BK);
    for (int k0 = ks + (int)threadIdx.x; k0 < ke; k0 += 512) {      // (two entries per thread and round trip: t0_theta_batch)
      float tb[2];
      t0_theta_batch<2>(a, s, k0, false, tb);
#pragma unroll
      for (int u = 0; u < 2; ++u) if (k0 + 256 * u < ke) a.w[s * a.Dp + k0 + 256 * u] = expf(-2.f * tb[u]);
    }
    __threadfence_block();
    __syncthreads();
  }
  gemm_body<64, 64, BK, true, true, true, true, true>(p, tile, batch, split, lds);
  STEP_SPAN_MARK(gemm, 10);
}

int launch_pro_kuu(const ProArgs& a, int npro, const NormArgs& n, const GemmParams& ps, int nbatch, hipStream_t st) {
  // (nbatch == 0: no product role -- the factorising workgroups of the next launch build their Gram matrices themselves, chol_gram.h)
  VARGP_REQUIRE(nbatch == 0 || (ps.splitk > 1 && ps.kscale == a.w && ps.ks_ld == a.Dp && gemm_vec_ok(ps)), "pro_kuu: not applicable");
  VARGP_REQUIRE(a.D <= kProKuuMaxD, "pro_kuu: D = %d out of range", a.D);
  if (prof_remembering()) {
    const ProArgs ac = a; const NormArgs nc = n; const GemmParams pc = ps;
    prof_remember("t0_pro_kuu", [=](hipStream_t s) { launch_pro_kuu(ac, npro, nc, pc, nbatch, s); });
  }
  ProfScope prof("t0_pro_kuu", st);
  static const int bk = [] { const char* e = getenv("VARGP_PRO_KUU_BK"); return e ? atoi(e) : 32; }();   // tuning aid
  GemmParams q = ps;
  q.xcd_remap = 0; q.group_m = 0;
  const int tiles = cdiv(q.M, 64) * cdiv(q.N, 64), ngemm = nbatch > 0 ? tiles * nbatch * q.splitk : 0, nnorm = n.nrow_blocks * a.S;
  const dim3 grid(ngemm + npro + nnorm);
  // norms_first: with the Lu role's dot products gone (ProArgs::su_in_chain) the prologue roles are short, and the norm role --
  // the longest of the small roles -- goes in front of them (VARGP_PRO_NORMS_FIRST overrides: tuning aid)
  static const int nf_env = [] { const char* e = getenv("VARGP_PRO_NORMS_FIRST"); return e ? atoi(e) : -1; }();
  const int nf = nf_env >= 0 ? nf_env : (a.su_in_chain ? 1 : 0);
  if (bk == 64) hipLaunchKernelGGL((t0_pro_kuu_kernel<64>), grid, dim3(256), 0, st, a, npro, n, nnorm, q, tiles, ngemm, nf);
  else hipLaunchKernelGGL((t0_pro_kuu_kernel<32>), grid, dim3(256), 0, st, a, npro, n, nnorm, q, tiles, ngemm, nf);
  return check_launch("pro_kuu");
}

// two problems, one launch; falls back to two launches when the pair kernel does not apply
int launch_gemm_pair(const GemmParams& p0, int nbatch0, const GemmParams& p1, int nbatch1, int transA, int transB,
                     bool rbf, hipStream_t st, const char* tag0, const char* tag1) {
  const bool ok = gemm_vec_ok(p0) && gemm_vec_ok(p1) && ((rbf && transA == 0 && transB == 1) || (!rbf && transA == 0 && transB == 0));
  if (!ok) {
    int rc = launch_gemm(p0, transA, transB, nbatch0, rbf, st, tag0);
    if (rc) return rc;
    return launch_gemm(p1, transA, transB, nbatch1, rbf, st, tag1);
  }
  if (prof_remembering() && strcmp(tag0, "replay") != 0) {
    const GemmParams c0 = p0, c1 = p1;
    prof_remember(tag0, [c0, c1, nbatch0, nbatch1, transA, transB, rbf](hipStream_t s) {
      launch_gemm_pair(c0, nbatch0, c1, nbatch1, transA, transB, rbf, s, "replay", "replay");
    });
  }
  ProfScope prof("gemm_pair", st);
  GemmPair pp;
  pp.p[0] = p0; pp.p[1] = p1;
  pp.p[0].splitk = pp.p[1].splitk = 1;
  pp.tiles[0] = cdiv(p0.M, 64) * cdiv(p0.N, 64);
  pp.tiles[1] = cdiv(p1.M, 64) * cdiv(p1.N, 64);
  pp.nwg0 = pp.tiles[0] * nbatch0;
  const int total = pp.nwg0 + pp.tiles[1] * nbatch1;
  if (rbf) hipLaunchKernelGGL((gemm_pair_kernel<64, 64, 64, true, true, true, true>), dim3(total), dim3(256), 0, st, pp);
  else hipLaunchKernelGGL((gemm_pair_kernel<64, 64, 64, true, false, true, false>), dim3(total), dim3(256), 0, st, pp);
  return check_launch("gemm_pair");
}

// factorisations (n in (50, 100], dense n x n matrices) + one RBF GEMM in one launch; false if the shapes do not
// qualify (the caller then launches them separately)
static int device_cu_count(int dev) {
  static std::atomic<int> cus[64] = {};
  int v = cus[dev].load(std::memory_order_acquire);
  if (v == 0) {
    hipDeviceProp_t prop;
    v = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    cus[dev].store(v, std::memory_order_release);
  }
  return v;
}
// CUs of the device the calling thread launches on (256 on MI355X)
static int current_cu_count() {
  int dev = 0;
  return (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) ? device_cu_count(dev) : 256;
}
int vargp_cu_count() { return current_cu_count(); }
static int launch_chol_rbf_gemm_impl(const float* A, float eps, float* L, float* T, int32_t* info, int nchol, int n,
                                     const GemmParams& p, int nbatch, hipStream_t st, const CholExtra& extra, const ZeroJobs& zero);
bool chol_rbf_gemm_applicable(int n, const GemmParams& p) { return n > 50 && n <= 100 && gemm_vec_ok(p); }
int launch_chol_rbf_gemm(const float* A, float eps, float* L, float* T, int32_t* info, int nchol, int n,
                         const GemmParams& p, int nbatch, hipStream_t st, const CholExtra* extra, const ZeroJobs* zero) {
  const CholExtra ex = extra ? *extra : CholExtra{};
  const ZeroJobs zj = zero ? *zero : ZeroJobs{};
  if (prof_remembering()) {
    const GemmParams pc = p;
    prof_remember("chol_rbf_gemm", [=](hipStream_t s) {
      launch_chol_rbf_gemm_impl(A, eps, L, T, info, nchol, n, pc, nbatch, s, ex, zj);
    });
  }
  return launch_chol_rbf_gemm_impl(A, eps, L, T, info, nchol, n, p, nbatch, st, ex, zj);
}
static int launch_chol_rbf_gemm_impl(const float* A, float eps, float* L, float* T, int32_t* info, int nchol, int n,
                                     const GemmParams& p, int nbatch, hipStream_t st, const CholExtra& extra, const ZeroJobs& zero) {
  // (nchol / nbatch are modified by the timing experiments below)
  ProfScope prof("chol_rbf_gemm", st);
  const int64_t nn = (int64_t)n * n;
  bool any_zero = false;
  for (int q = 0; q < kZeroJobs; ++q) any_zero = any_zero || zero.j[q].p != nullptr;
  CholArgs c{A, n, nn, eps, L, n, nn, T, n, nn, info, n, nchol, extra, zero, any_zero ? 64 : 0};
  GemmParams q = p;
  q.splitk = 1;
  // GEMM tile: 64x64x64 normally; 128x64x32 when that lets the GEMM finish in ONE round on the CUs the
  // factorisations leave free (the BASELINE shape: 384 tiles on 216 CUs would need two rounds, 192 need one)
  static const int tile_force = [] { const char* e = getenv("VARGP_MERGED_TILE"); return e ? atoi(e) : 0; }();   // tuning aid
  const int t64 = cdiv(p.M, 64) * cdiv(p.N, 64), t128 = cdiv(p.M, 128) * cdiv(p.N, 64);
  const int free_cus = current_cu_count() - nchol;
  // (with the 8-wave GEMM role below -- pre-scaled operand, fp32 chains, n > 64 -- the 128 x 64 tile is the better one whenever
  // 64 x 64 tiles need more than one round: S = 16 step 869 -> 831 us, S = 8 477 -> 474, S >= 32 the same)
  static const int f32_env = [] { const char* e = getenv("VARGP_CHOL_F32"); return e ? atoi(e) : kCholF32Default; }();
  const bool wide_ok = p.kscale == nullptr && f32_env && n > 64;
  const bool big = nbatch == 0 ? false
                   : tile_force ? tile_force == 2
                                : (t64 * nbatch > free_cus && (wide_ok || t128 * nbatch <= free_cus));
  const int tiles = big ? t128 : t64;
  // VARGP_EXP_MERGED (timing only, wrong results): 1 = the factorisations alone (no GEMM tiles), 2 = the GEMM tiles alone
  static const int exp_role = [] { const char* e = getenv("VARGP_EXP_MERGED"); return e ? atoi(e) : 0; }();
  if (exp_role == 2) { c.nchol = 0; nchol = 0; }
  if (exp_role == 1) nbatch = 0;
  // the zero-fill role: as many workgroups as CUs are left over (every workgroup of this launch has a CU to itself: registers), so
  // that they start with the launch -- 64 of them behind 232 others waited for the first chain or tile to end and WERE the end of
  // the launch (span 37.7 us with the chains done at 31.4 and the tiles at 31)
  if (any_zero && nbatch > 0) {
    const int spare = current_cu_count() - nchol - tiles * nbatch;
    if (spare >= 8) c.nzero = spare < 64 ? spare : 64;
  }
  const int total = nchol + tiles * nbatch + c.nzero;
  // The kernel needs 68 KB of LDS, so two workgroups fit a CU.  While the GEMM is small enough to finish under the
  // factorisations anyway (the BASELINE shapes), reserving unused dynamic LDS keeps every factorising CU to itself --
  // a co-resident GEMM workgroup competes for its issue slots and stretches the pivot chain (68 -> 77 us measured);
  // with many samples the GEMM dominates and wants both slots (S = 64: 950 -> 676 us).
  static const int pad_force = [] { const char* e = getenv("VARGP_MERGED_PAD"); return e ? atoi(e) : -1; }();   // tuning aid (KB)
  // (no product in the launch -- many hyper-samples, vargp_elbo_t0_fwd -- and more chains than CUs: two chains per CU)
  const unsigned pad = pad_force >= 0 ? (unsigned)pad_force * 1024u
                       : (nbatch == 0 ? (nchol > current_cu_count() ? 0u : 24u * 1024u)
                                      : (tiles * nbatch <= 1024 ? (big ? 40u : 24u) * 1024u : 0u));
  // arithmetic of the pivot chains: the reference's own fp32 (default, kCholF32Default = 1: torch.cholesky is fp32) or fp64
  // (VARGP_CHOL_F32=0; chol_small3.h).  The stand-alone factorisation (vargp_chol_inv_fwd: predict, the composed path) is fp64.
  // kscale == NULL: the caller's B operand is pre-scaled (x o 1/sigma^2, written once per hyper-sample by the norm role) and the
  // main loop carries no scale loads and multiplies
  static const int exp_unscaled = [] { const char* e = getenv("VARGP_EXP_UNSCALED"); return e ? atoi(e) : 0; }();   // timing only
  const bool scaled = q.kscale != nullptr && !exp_unscaled;
  // 8-wave GEMM role (two waves per SIMD on the same 128 x 64 x 32 tile; the factorising workgroups use their first four waves):
  // pays since the launch is bound by its GEMM role (the fp32 chains end 6 us before the K_uf tiles)
  static const int nt_env = [] { const char* e = getenv("VARGP_MERGED_NT"); return e ? atoi(e) : 512; }();   // tuning aid
  const bool wide = big && !scaled && nt_env == 512 && f32_env && n > 64;
#define VARGP_MERGED(KC, SETS, R)                                                                                                  \
  do {                                                                                                                               \
    if (big && scaled) hipLaunchKernelGGL((chol_rbf_gemm_kernel<KC, SETS, 128, 32, true, R>), dim3(total), dim3(256), pad, st, c, q, tiles);  \
    else if (big) hipLaunchKernelGGL((chol_rbf_gemm_kernel<KC, SETS, 128, 32, false, R>), dim3(total), dim3(256), pad, st, c, q, tiles);     \
    else if (scaled) hipLaunchKernelGGL((chol_rbf_gemm_kernel<KC, SETS, 64, 64, true, R>), dim3(total), dim3(256), pad, st, c, q, tiles);    \
    else hipLaunchKernelGGL((chol_rbf_gemm_kernel<KC, SETS, 64, 64, false, R>), dim3(total), dim3(256), pad, st, c, q, tiles);              \
  } while (0)
  if (wide) hipLaunchKernelGGL((chol_rbf_gemm_kernel<25, 2, 128, 32, false, float, 512>), dim3(total), dim3(512), pad, st, c, q, tiles);
  else if (f32_env) { if (n <= 64) VARGP_MERGED(16, 1, float); else VARGP_MERGED(25, 2, float); }
  else { if (n <= 64) VARGP_MERGED(16, 1, double); else VARGP_MERGED(25, 2, double); }
#undef VARGP_MERGED
  return check_launch("chol_rbf_gemm");
}

int launch_chol_rbf_gemm_ld(const float* A, int lda, int64_t sA, float eps, float* L, int ldl, int64_t sL, float* T, int ldt,
                            int64_t sT, int32_t* info, int nchol, int n, const GemmParams& p, int nbatch, hipStream_t st,
                            bool chain_f32, const ZeroJobs* zero) {
  const ZeroJobs zj = zero ? *zero : ZeroJobs{};
  if (prof_remembering()) {
    const GemmParams pc = p;
    prof_remember("chol_rbf_gemm", [=](hipStream_t s) {
      launch_chol_rbf_gemm_ld(A, lda, sA, eps, L, ldl, sL, T, ldt, sT, info, nchol, n, pc, nbatch, s, chain_f32, &zj);
    });
  }
  ProfScope prof("chol_rbf_gemm", st);
  bool any_zero = false;
  for (int q = 0; q < kZeroJobs; ++q) any_zero = any_zero || zj.j[q].p != nullptr;
  // (zero-fills of the blocked driver, chol.hip: 2 x 64 MB at Permuted-MNIST t = 1 -- as many workgroups as CUs)
  CholArgs c{A, lda, sA, eps, L, ldl, sL, T, ldt, sT, info, n, nchol, CholExtra{}, zj, any_zero ? current_cu_count() : 0};
  GemmParams q = p;
  q.splitk = 1;
  const int t64 = cdiv(p.M, 64) * cdiv(p.N, 64), t128 = cdiv(p.M, 128) * cdiv(p.N, 64);
  const int free_cus = current_cu_count() - nchol;
  const bool big = t64 * nbatch > free_cus && (t128 * nbatch <= free_cus || t64 * nbatch > 1024);
  const int tiles = big ? t128 : t64;
  const int total = nchol + tiles * nbatch + c.nzero;
  // factorising CUs exclusive (see launch_chol_rbf_gemm_impl) only while the GEMM fits one round on the other CUs
  const unsigned pad = tiles * nbatch <= free_cus ? (big ? 40u : 24u) * 1024u : 0u;
  const bool scaled = p.kscale != nullptr;      // NULL: the caller's B operand is pre-scaled (rbf_prep_norm_launch: ys)
#define VARGP_MERGED(KC, SETS, R)                                                                                           \
  do {                                                                                                                        \
    if (big && scaled) hipLaunchKernelGGL((chol_rbf_gemm_kernel<KC, SETS, 128, 32, true, R>), dim3(total), dim3(256), pad, st, c, q, tiles);   \
    else if (big) hipLaunchKernelGGL((chol_rbf_gemm_kernel<KC, SETS, 128, 32, false, R>), dim3(total), dim3(256), pad, st, c, q, tiles);      \
    else if (scaled) hipLaunchKernelGGL((chol_rbf_gemm_kernel<KC, SETS, 64, 64, true, R>), dim3(total), dim3(256), pad, st, c, q, tiles);     \
    else hipLaunchKernelGGL((chol_rbf_gemm_kernel<KC, SETS, 64, 64, false, R>), dim3(total), dim3(256), pad, st, c, q, tiles);                \
  } while (0)
  // pivot chains of the diagonal blocks in the reference's own fp32 (torch.cholesky, gp_utils.py:5-11), as in the first-task
  // launch above (VARGP_CHOL_F32=0: fp64) -- when the caller asks for it: the TRAINING forward of the block program does
  // (chain_f32); gradient-free evaluation (predict, accuracy sweeps) and the stand-alone factorisation stay fp64.
  static const int f32_env = [] { const char* e = getenv("VARGP_CHOL_F32"); return e ? atoi(e) : kCholF32Default; }();
  if (f32_env && chain_f32) { if (n <= 64) VARGP_MERGED(16, 1, float); else VARGP_MERGED(25, 2, float); }
  else
  { if (n <= 64) VARGP_MERGED(16, 1, double); else VARGP_MERGED(25, 2, double); }
#undef VARGP_MERGED
  return check_launch("chol_rbf_gemm");
}

// chains of nchol diagonal blocks (n in (50, 100], fp64) || the plain NN product p over nbatch matrices (see chol_nn_gemm_kernel)
bool chol_nn_gemm_applicable(int n, const GemmParams& p) {
  return n > 50 && n <= 100 && gemm_vec_ok(p) && p.splitk <= 1 && p.M > 0 && p.N > 0 && p.K > 0;
}
int launch_chol_nn_gemm(const float* A, int lda, int64_t sA, float eps, float* L, int ldl, int64_t sL, float* T, int ldt, int64_t sT,
                        int32_t* info, int info_base, int nchol, int n, const GemmParams& p, int nbatch, hipStream_t st,
                        const ZeroJobs* zero) {
  ProfScope prof("chol_nn_gemm", st);
  const ZeroJobs zj = zero ? *zero : ZeroJobs{};
  bool any_zero = false;
  for (int q = 0; q < kZeroJobs; ++q) any_zero = any_zero || zj.j[q].p != nullptr;
  // (zero-fills of the blocked driver: up to 2 x 160 MB at n = 2048 x 10 -- every CU the chains leave free takes part)
  CholArgs c{A, lda, sA, eps, L, ldl, sL, T, ldt, sT, info, n, nchol, CholExtra{}, zj, any_zero ? current_cu_count() - nchol : 0};
  GemmParams q = p;
  q.splitk = 1; q.nofast = 0; q.group_m = 0; q.xcd_remap = 1;
  const int tiles = cdiv(p.M, 64) * cdiv(p.N, 64);     // (p.M == 0: no product, chains (+ zero-fills) only)
  const int total = nchol + tiles * nbatch + c.nzero;
  if (n <= 64) hipLaunchKernelGGL((chol_nn_gemm_kernel<16, 1, double>), dim3(total), dim3(256), 0, st, c, q, tiles, info_base);
  else hipLaunchKernelGGL((chol_nn_gemm_kernel<25, 2, double>), dim3(total), dim3(256), 0, st, c, q, tiles, info_base);
  return check_launch("chol_nn_gemm");
}

// two plain products with their own transposition flags in one launch; the layout pairs the ELBO program uses are
// instantiated, anything else (or unaligned operands) falls back to two launches
int launch_gemm_pair2(const GemmParams& p0, int tA0, int tB0, int nbatch0, const GemmParams& p1, int tA1, int tB1,
                      int nbatch1, hipStream_t st, const char* tag) {
  const bool a0 = tA0 == 0, b0 = tB0 == 1, a1 = tA1 == 0, b1 = tB1 == 1;   // operand K-contiguous?
  const bool nt_nn = a0 && b0 && a1 && !b1, nt_tn = a0 && b0 && !a1 && !b1;
  const bool nn_nn = a0 && !b0 && a1 && !b1, tn_tn = !a0 && !b0 && !a1 && !b1;
  if (!(gemm_vec_ok(p0) && gemm_vec_ok(p1) && (nt_nn || nt_tn || nn_nn || tn_tn))) {
    int rc = launch_gemm(p0, tA0, tB0, nbatch0, false, st, tag);
    if (rc) return rc;
    return launch_gemm(p1, tA1, tB1, nbatch1, false, st, tag);
  }
  if (prof_remembering() && strcmp(tag, "replay") != 0) {
    const GemmParams c0 = p0, c1 = p1;
    prof_remember(tag, [=](hipStream_t s) { launch_gemm_pair2(c0, tA0, tB0, nbatch0, c1, tA1, tB1, nbatch1, s, "replay"); });
  }
  ProfScope prof(tag, st);
  GemmPair pp;
  pp.p[0] = p0; pp.p[1] = p1;
  for (int i = 0; i < 2; ++i) {   // splitk > 1: atomic accumulation into a pre-zeroed C (no D, no scratch)
    if (pp.p[i].splitk < 1) pp.p[i].splitk = 1;
    pp.p[i].sSplit = 0;
  }
  pp.tiles[0] = cdiv(p0.M, 64) * cdiv(p0.N, 64);
  pp.tiles[1] = cdiv(p1.M, 64) * cdiv(p1.N, 64);
  pp.per_split[0] = pp.tiles[0] * nbatch0;
  pp.per_split[1] = pp.tiles[1] * nbatch1;
  pp.nwg0 = pp.per_split[0] * pp.p[0].splitk;
  const int total = pp.nwg0 + pp.per_split[1] * pp.p[1].splitk;
  if (nt_nn) hipLaunchKernelGGL((gemm_pair2_kernel<true, true, true, false>), dim3(total), dim3(256), 0, st, pp);
  else if (nt_tn) hipLaunchKernelGGL((gemm_pair2_kernel<true, true, false, false>), dim3(total), dim3(256), 0, st, pp);
  else if (nn_nn) hipLaunchKernelGGL((gemm_pair2_kernel<true, false, true, false>), dim3(total), dim3(256), 0, st, pp);
  else hipLaunchKernelGGL((gemm_pair2_kernel<false, false, false, false>), dim3(total), dim3(256), 0, st, pp);
  return check_launch("gemm_pair2");
}

// what gemm_persist_body needs of a plain product (BK-deep slabs, A K-contiguous or not, B likewise)
static bool gemm_persist_ok(const GemmParams& p, int BK, bool AKC, bool BKC) {
  const int64_t extA = AKC ? ((int64_t)(p.M - 1) * p.lda + p.K) : ((int64_t)(p.K - 1) * p.lda + p.M);
  const int64_t extB = BKC ? ((int64_t)(p.N - 1) * p.ldb + p.K) : ((int64_t)(p.K - 1) * p.ldb + p.N);
  return gemm_vec_ok(p) && p.K >= 4 * BK && p.K % BK == 0 && !p.D && p.triA == 0 && p.triB == 0 && p.triC == 0 && !p.symout &&
         p.splitk <= 1 && (AKC || (p.M % 4 == 0 && p.M >= 4)) && (BKC || (p.N % 4 == 0 && p.N >= 4)) &&
         extA < (1ll << 29) && extB < (1ll << 29);
}

int launch_bwdmat_gemm(const BwdMatArgs& a, int first, int nmat, const GemmParams& p, int nbatch, hipStream_t st,
                       const char* tag, int* queue) {
  VARGP_REQUIRE(gemm_vec_ok(p), "bwdmat_gemm: the product's operands must be 16-byte aligned with strides % 4 == 0");
  VARGP_REQUIRE(a.M <= kBmKP && a.M >= 4 && (a.M % 4) == 0 && (a.LD % 4) == 0, "bwdmat_gemm: M = %d out of range", a.M);
  static_assert(kBwdMatLdsBytes >= sizeof(float) * gemm_lds_floats<64, 64, 64, true, false>(), "LDS of the GEMM role");
  if (prof_remembering() && strcmp(tag, "replay") != 0) {
    const GemmParams pc = p;
    const BwdMatArgs ac = a;
    prof_remember(tag, [=](hipStream_t s) { launch_bwdmat_gemm(ac, first, nmat, pc, nbatch, s, "replay", nullptr); });
  }
  // hipFuncAttributeMaxDynamicSharedMemorySize is per device
  static std::atomic<unsigned> attr_mask[2] = {};
  int dev = 0;
  VARGP_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64, "bwdmat_gemm: hipGetDevice failed");
  if (!((attr_mask[dev >> 5].load(std::memory_order_acquire) >> (dev & 31)) & 1u)) {
    VARGP_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(t0_bwdmat_gemm_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBwdMatLdsBytes) == hipSuccess,
                  "bwdmat_gemm: cannot reserve %zu bytes of LDS", kBwdMatLdsBytes);
    attr_mask[dev >> 5].fetch_or(1u << (dev & 31), std::memory_order_release);
  }
  ProfScope prof(tag, st);
  GemmParams q = p;
  q.splitk = 1; q.nofast = 0; q.group_m = 0; q.xcd_remap = 1;
  const int tiles = cdiv(p.M, 64) * cdiv(p.N, 64);
  // tuning aid (timing of one role alone; results are then incomplete): VARGP_EXP_BWDMAT = 1: matrices only, 2: product only
  static const int exp_role = [] { const char* e = getenv("VARGP_EXP_BWDMAT"); return e ? atoi(e) : 0; }();
  if (exp_role == 1) nbatch = 0;
  if (exp_role == 2) nmat = 0;
  // the product's tiles: one workgroup each, or -- more tiles than free CUs -- one persistent workgroup per free CU (every
  // workgroup of this launch has a CU to itself: the chain's LDS).  VARGP_GEMM_PERSIST (tuning aid): 0 off, 2 / 3 force the
  // work queue / the static lists
  static const int persist_env = [] { const char* e = getenv("VARGP_GEMM_PERSIST"); return e ? atoi(e) : 1; }();   // tuning aid
  const int total = tiles * nbatch;
  const int free_cus = device_cu_count(dev) - nmat;
  const bool persist = persist_env && total > free_cus && free_cus >= 8 && gemm_persist_ok(q, 64, true, false);
  // few tiles per CU (the chains take about as long as the product): static tile lists; many: the work queue, which the chains'
  // workgroups join when they are done (needs the caller's 8 zeroed counters)
  const bool use_queue = persist && queue && (persist_env == 2 || (persist_env != 3 && total > 4 * free_cus));
  const int ngemm = persist ? free_cus : total;
  hipLaunchKernelGGL(t0_bwdmat_gemm_kernel, dim3(nmat + ngemm), dim3(256), kBwdMatLdsBytes, st, a, first, nmat, q, tiles,
                     use_queue ? queue : (int*)nullptr, total, persist ? 1 : 0);
  return check_launch("bwdmat_gemm");
}

static int g_tile_force = 0;   // vargp_tune_gemm_tile (measurement only)

int launch_gemm(const GemmParams& p, int transA, int transB, int nbatch, bool rbf, hipStream_t st, const char* tag) {
  if (p.M <= 0 || p.N <= 0 || nbatch <= 0) return VARGP_OK;
  if (prof_remembering() && strcmp(tag, "replay") != 0) {
    const GemmParams pc = p;
    prof_remember(tag, [pc, transA, transB, nbatch, rbf](hipStream_t s) { launch_gemm(pc, transA, transB, nbatch, rbf, s, "replay"); });
  }
  static const int nofast = [] { const char* e = getenv("VARGP_GEMM_NOFAST"); return e ? atoi(e) : 0; }();   // tuning aid
  const_cast<GemmParams&>(p).nofast = nofast;
  ProfScope prof(tag, st);
  VARGP_REQUIRE(nbatch <= 65535, "bgemm: batch %d exceeds 65535", nbatch);
  {
    // XCD-compact tile ranges pay while one XCD's share of the operands fits its 4 MB L2; on big grids (the
    // M = 2048 predictive sweep: 25 MB of B per XCD) the plain map, which spreads neighbours over the XCDs, is faster
    static const int xcd_force = [] { const char* e = getenv("VARGP_GEMM_XCD"); return e ? atoi(e) : -1; }();   // tuning aid
    const int64_t wgs = (int64_t)cdiv(p.M, 64) * cdiv(p.N, 64) * nbatch;
    // ... except with triangular operands / results, whose tiles differ in length: there the compact ranges (whole
    // matrices per XCD, long tiles first) balance much better at any size (1000 x 512 x 1000, batch 100: 633 -> 507 us;
    // Permuted-MNIST t=4 step 55 -> 70 steps/s).  Dense products are within 2 % either way (4096^3, 1000^3 batch 100;
    // a grouped 2-D tile order changed nothing for them: they are not limited by L2 misses).
    // (batches of matrices up to ~2000 x 2000; the M = 2048 x 8192 products of the N = 1e6 sweep, 10 matrices on 8 XCDs,
    //  are 3 % faster with the plain map)
    const bool tri_any = p.triA != 0 || p.triB != 0 || p.triC != 0;
    const bool tri_batch = tri_any && (int64_t)cdiv(p.M, 64) * cdiv(p.N, 64) <= 2048;
    // Large dense products: compact ranges AND a grouped (8 tile rows at a time) order inside them -- same speed within
    // 2 %, but a fraction of the L2 misses.
    static const int grp = [] { const char* e = getenv("VARGP_GEMM_GROUP"); return e ? atoi(e) : 8; }();   // tuning aid
    const bool grouped = grp > 1 && wgs > 4096 && !tri_any;
    const_cast<GemmParams&>(p).group_m = grouped ? grp : 0;
    const_cast<GemmParams&>(p).xcd_remap = xcd_force >= 0 ? xcd_force : ((wgs <= 4096 || tri_batch || grouped) ? 1 : 0);
  }
  bool vec = aligned16(p.A) && aligned16(p.B) && (p.lda % 4 == 0) && (p.ldb % 4 == 0);
  for (int i = 0; i < 3; ++i) vec = vec && (p.sA[i] % 4 == 0) && (p.sB[i] % 4 == 0);
  const int64_t t128 = (int64_t)cdiv(p.M, 128) * cdiv(p.N, 128) * nbatch;
  const int64_t t12864 = (int64_t)cdiv(p.M, 128) * cdiv(p.N, 64) * nbatch;
  static const int force_env = [] { const char* e = getenv("VARGP_GEMM_TILE"); return e ? atoi(e) : 0; }();   // tuning aid
  const int force = g_tile_force ? g_tile_force : (force_env ? force_env : p.tile);
  // Tile choice, from the sweep `tests/native/bench_kernels tiles` on MI355X (mid-size batched products of the ELBO
  // programs): 128x64x32 is the robust mid-size shape; 64x64x64 when it pads M less and K ranges are clipped by a
  // triangular operand (finer clipping, better balance: 400x512x400 b100 140 vs 182/192 us) or when the problem is small
  // (more workgroups); 128x128x16 only for large square-ish problems (>= 1024 both ways), where it reaches 80-90 % of peak.
  static const int tric64 = [] { const char* e = getenv("VARGP_GEMM_TRIC64"); return e ? atoi(e) : 1; }();   // tuning aid
  const bool tri = p.triA != 0 || p.triB != 0 || (tric64 && p.triC != 0);
  const bool pad64_less = round_up(p.M, 64) < round_up(p.M, 128);
  if (force == 1) dispatch_tile<128, 128, 16>(p, transA, transB, nbatch, rbf, vec, st);
  else if (force == 2) dispatch_tile<128, 64, 32>(p, transA, transB, nbatch, rbf, vec, st);
  else if (force == 3) dispatch_tile<64, 64, 64>(p, transA, transB, nbatch, rbf, vec, st);
  // (a transposed A, e.g. V2 = T^T P, is the exception: 400 x 512 x 400 b100 123 vs 136 us with 128 x 64 x 32)
  else if (t12864 >= 384 && tri && pad64_less && transA == 0) dispatch_tile<64, 64, 64>(p, transA, transB, nbatch, rbf, vec, st);
  else if (t128 >= 512 && !tri && p.M >= 1024 && p.N >= 1024) dispatch_tile<128, 128, 16>(p, transA, transB, nbatch, rbf, vec, st);
  else if (t12864 >= 384 && p.M > 64) dispatch_tile<128, 64, 32>(p, transA, transB, nbatch, rbf, vec, st);
  else dispatch_tile<64, 64, 64>(p, transA, transB, nbatch, rbf, vec, st);
  return check_launch("bgemm");
}

}  // namespace vargp

extern "C" int vargp_tune_gemm_tile(int tile) {
  vargp::g_tile_force = (tile >= 0 && tile <= 3) ? tile : 0;
  return VARGP_OK;
}

extern "C" int vargp_bgemm(const vargp_gemm_desc* d, vargp_stream_t stream) {
  using namespace vargp;
  VARGP_REQUIRE(d != nullptr, "bgemm: null descriptor");
  VARGP_REQUIRE(d->A && d->B && d->C, "bgemm: null operand");
  VARGP_REQUIRE(d->M >= 0 && d->N >= 0 && d->K >= 0, "bgemm: negative dims");
  GemmParams p{};
  p.A = d->A; p.B = d->B; p.C = d->C; p.D = d->D;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc; p.ldd = d->ldd;
  const int nb0 = d->nb[0] > 0 ? d->nb[0] : 1;
  p.nb1 = d->nb[1] > 0 ? d->nb[1] : 1;
  p.nb2 = d->nb[2] > 0 ? d->nb[2] : 1;
  for (int i = 0; i < 3; ++i) { p.sA[i] = d->sA[i]; p.sB[i] = d->sB[i]; p.sC[i] = d->sC[i]; p.sD[i] = d->sD[i]; }
  p.alpha = d->alpha; p.beta = d->D ? d->beta : 0.f;
  p.triA = d->triA; p.triB = d->triB; p.triC = d->triC;
  return launch_gemm(p, d->transA, d->transB, nb0 * p.nb1 * p.nb2, false, as_stream(stream));
}
