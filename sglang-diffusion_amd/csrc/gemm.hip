// MFMA GEMMs for gfx950:  out[M,N] = epilogue( X[M,K] . W[N,K]^T ).
//
// One kernel template serves the bf16 linear (v_mfma_f32_32x32x16_bf16) and
// the W8A8 fp8 linear (v_mfma_scale_f32_32x32x64_f8f6f4 with unit block
// scales: the only fp8 MFMA form that runs at twice the bf16 rate on CDNA4;
// per-token x per-channel scales are applied once, in the epilogue).
//
// Data movement.  Both operands are K-contiguous, so an operand tile is
// [rows][128 bytes] for either dtype (64 bf16 or 128 fp8 along K).  Tiles go
// HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR round trip), double
// buffered, one barrier per K tile.  The LDS image is written lane-linearly
// (that is what LDS-DMA does), so the bank swizzle is applied to the per-lane
// SOURCE address and again on the ds_read_b128 side:
//      physical 16-B chunk = logical chunk ^ ((row >> 1) & 7).
// With 128-B rows a 256-B bank row holds two tile rows; the XOR spreads the
// 16 lanes that ds_read_b128 services per cycle over all 16 chunk slots.
//
// MFMA orientation.  A-operand = W fragment, B-operand = X fragment, so the
// 32x32 accumulator has the token (m) on the lane and 4-element runs of n in
// registers: after one permlane32_swap per register pair every lane owns 8
// consecutive output channels of one token and stores 16 bytes.
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <type_traits>

#include "common.h"

namespace {

struct GemmParams {
  const unsigned char* a;
  const unsigned char* w;
  const float* sa;
  const float* sb;
  const unsigned short* bias;
  unsigned short* out;
  const unsigned short* residual;
  const unsigned short* gate;
  int M, N, ktiles;
  int64_t lda_b, ldw_b;  // row strides in BYTES
  int64_t ldo, ldg;      // element strides
  int64_t gate_rows;
  mi355_rowmap a_map, o_map;
  int act;
  int f16_begin, f16_end;  // output columns stored as half
  unsigned long long* f16_counters;  // range counters of those columns (see common.h)
  // block-scaled variant
  int64_t sa_sm, sa_sk, sb_sk, sb_sn;
  int group;  // M tiles per XCD chunk of the tile walk
  int out_f16;  // host side only: picks the instantiation whose output and bias are IEEE half
  int split_from;  // ring kernel: tiles from this id on are run as two 128-channel halves
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;


// ---- epilogue (N % 8 == 0 is a launch precondition) ---------------------------------
// Runs in rounds of (32 rows) x (64 channels) per wave through a 4-KiB LDS scratch so that
// global stores are whole 128-B lines: (1) the wave turns one round of accumulators into
// bf16 (scales, bias, activation) and writes it as a [32][64] image; (2) it reads the image
// back row-major, 16 B per lane, applies gate/residual and stores 128 contiguous bytes per
// row.  LDS operations of one wave execute in order, so the rounds need no barrier.
// sb_l / sa_l / bias_l: the tile's per-channel scales, per-token scales and bias in LDS.
// OUT16: the whole output (and the bias) is IEEE half, rounded once from fp32 -- out_dtype
// float16 of the reference's fp8 GEMM ops; no gate / residual / half-column range in that mode.
template <int DT, int WAVES_M, int WAVES_N, int TM, int TN, bool BLOCKSCALE, bool OUT16 = false>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x16 (&acc)[TN][TM],
                                              unsigned char* scratch, const float* sb_l,
                                              const float* sa_l, const unsigned short* bias_l,
                                              int m0, int n0, int wave, int lane) {
  static_assert(TN % 2 == 0, "rounds cover two 32-channel blocks");
  const int wm = wave % WAVES_M;
  const int wn = wave / WAVES_M;
  const int r = lane & 31;
  const int h = lane >> 5;
  const int lc = lane & 7, lr = lane >> 3;
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    float sam = 1.f;
    if constexpr (DT == 1 && !BLOCKSCALE) sam = sa_l[wm * TM * 32 + j * 32 + r];
    const int mw = m0 + wm * TM * 32 + j * 32;
#pragma unroll
    for (int ip = 0; ip < TN / 2; ++ip) {
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) {
        const int i = 2 * ip + i2;
        // wave-uniform: the half-format column range is 32-aligned
        const int nblk = n0 + wn * TN * 32 + i * 32;
        const bool as_f16 = nblk >= p.f16_begin && nblk < p.f16_end;
        unsigned int pk[4][2];
        float amax = 0.f;  // largest |value| of this 32-row x 32-channel block (half columns)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          // per-channel scale and bias of this lane's run of 4 channels
          const int nl = wn * TN * 32 + i * 32 + 8 * g + 4 * h;
          f32x4 sbv = f32x4{1.f, 1.f, 1.f, 1.f};
          if constexpr (DT == 1 && !BLOCKSCALE) sbv = *reinterpret_cast<const f32x4*>(sb_l + nl);
          const u32x2 bb = *reinterpret_cast<const u32x2*>(bias_l + nl);
          f32x4 bsv;
          if constexpr (OUT16) {
            auto hf = [](unsigned int bits) {
              return (float)__builtin_bit_cast(_Float16, (unsigned short)bits);
            };
            bsv = f32x4{hf(bb[0] & 0xffffu), hf(bb[0] >> 16), hf(bb[1] & 0xffffu), hf(bb[1] >> 16)};
          } else {
            bsv = f32x4{__uint_as_float(bb[0] << 16), __uint_as_float(bb[0] & 0xffff0000u),
                        __uint_as_float(bb[1] << 16), __uint_as_float(bb[1] & 0xffff0000u)};
          }
          float y[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = acc[i][j][4 * g + e];
            // the block-scaled linear casts the product to bf16 and THEN adds the bias
            // (python/sglang/srt/layers/quantization/fp8_utils.py:546-548)
            if constexpr (BLOCKSCALE && !OUT16) v = rbf(v);
            if constexpr (BLOCKSCALE && OUT16) v = (float)(_Float16)v;
            // per-token x per-channel scales and the bias: one multiply, one fused multiply-add
            if constexpr (DT == 1 && !BLOCKSCALE)
              v = fmaf(v * sam, sbv[e], bsv[e]);
            else
              v += bsv[e];
            y[e] = v;
          }
          if (p.act == MI355_ACT_GELU_TANH) {
            const f32x2 g0 = gelu_tanh_f2(f32x2{rbf(y[0]), rbf(y[1])});
            const f32x2 g1 = gelu_tanh_f2(f32x2{rbf(y[2]), rbf(y[3])});
            y[0] = g0[0];
            y[1] = g0[1];
            y[2] = g1[0];
            y[3] = g1[1];
          }
          if constexpr (OUT16) {
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            pk[g][0] = __builtin_bit_cast(unsigned int, h2{(_Float16)y[0], (_Float16)y[1]});
            pk[g][1] = __builtin_bit_cast(unsigned int, h2{(_Float16)y[2], (_Float16)y[3]});
          } else if (as_f16) {
            pk[g][0] = pack_bf2_as_f16(y[0], y[1]);
            pk[g][1] = pack_bf2_as_f16(y[2], y[3]);
#pragma unroll
            for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fabsf(y[e]));
          } else {
            pk[g][0] = pack_bf2(y[0], y[1]);
            pk[g][1] = pack_bf2(y[2], y[3]);
          }
        }
        if constexpr (!OUT16)
          if (as_f16) note_f16_block(p.f16_counters, group_max<64>(amax), lane);
        // widen: lanes l and l+32 exchange so each owns 8 consecutive channels
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            auto sw = __builtin_amdgcn_permlane32_swap(pk[g][d], pk[g + 1][d], false, false);
            pk[g][d] = sw[0];
            pk[g + 1][d] = sw[1];
          }
          const int chunk = i2 * 4 + g + h;
          *reinterpret_cast<u32x4*>(scratch + r * 128 + ((chunk ^ (r & 7)) << 4)) =
              u32x4{pk[g][0], pk[g][1], pk[g + 1][0], pk[g + 1][1]};
        }
      }
      const int n8 = n0 + wn * TN * 32 + ip * 64 + lc * 8;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + lr;
        const int m = mw + row;
        u32x4 o = *reinterpret_cast<const u32x4*>(scratch + row * 128 + ((lc ^ (row & 7)) << 4));
        if (m < p.M && n8 < p.N) {
          const int64_t orow =
              map_row32(m, (int)p.o_map.rows_per_batch, p.o_map.batch_rows, p.o_map.row_offset) *
              p.ldo;
          if (p.gate) {
            const int64_t grow = (int64_t)((unsigned)m / (unsigned)p.gate_rows) * p.ldg;
            const u32x4 gv = *reinterpret_cast<const u32x4*>(p.gate + grow + n8);
            const u32x4 rv = *reinterpret_cast<const u32x4*>(p.residual + orow + n8);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float y0 = __uint_as_float(o[e] << 16), y1 = __uint_as_float(o[e] & 0xffff0000u);
              const float g0 = __uint_as_float(gv[e] << 16),
                          g1 = __uint_as_float(gv[e] & 0xffff0000u);
              const float r0 = __uint_as_float(rv[e] << 16),
                          r1 = __uint_as_float(rv[e] & 0xffff0000u);
              o[e] = pack_bf2(r0 + rbf(g0 * y0), r1 + rbf(g1 * y1));
            }
          }
          __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(p.out + orow + n8));
        }
      }
    }
  }
}

// The same epilogue for accumulators of the 16x16 MFMA shapes (v_mfma_f32_16x16x32_bf16,
// v_mfma_scale_f32_16x16x128_f8f6f4).  acc[ia][jb] is the 16-channel x 16-token block
// (ia, jb) of the wave's tile: lane l holds token jb * 16 + (l & 15) and the four channels
// ia * 16 + 4 * (l >> 4) .. + 3.  Element arithmetic, rounding points and the store phase are
// those of gemm_epilogue; only the way a round's [32][64] image is written differs (8 bytes
// per lane and block instead of 16 after an exchange).
//
// This code runs with the matrix pipe idle (4-5 us of a 40-us tile at K = 3072), and what it
// costs is instruction issue and exposed LDS latency, not bytes: without its global stores it
// takes the same time.  Hence MODE: what a launch does NOT use (activation, half-format column
// range, gate / residual) is compiled out of its copy, so that a round is one basic block in
// which the table reads can be scheduled ahead of their use; and the addresses of a round's LDS
// image are formed once per tile (the chunk swizzle is an XOR with a per-lane constant, so a
// block's address is one of four per-lane values plus an immediate).
enum : int { EPI_GENERAL = 0, EPI_PLAIN = 1, EPI_ACT = 2, EPI_F16 = 3, EPI_GATE = 4 };

template <int DT, int WAVES_M, int WAVES_N, int TM, int TN, int MODE>
__device__ __forceinline__ void gemm_epilogue16_mode(const GemmParams& p,
                                                     f32x4 (&acc)[TN * 2][TM * 2],
                                                     unsigned char* scratch, const float* sb_l,
                                                     const float* sa_l, const unsigned short* bias_l,
                                                     int m0, int n0, int wave, int lane) {
  static_assert(TN % 2 == 0, "rounds cover two 32-channel blocks");
  constexpr bool ANY = MODE == EPI_GENERAL;
  const bool gelu = ANY ? p.act == MI355_ACT_GELU_TANH : MODE == EPI_ACT;
  const bool gated = ANY ? p.gate != nullptr : MODE == EPI_GATE;
  const int wm = wave % WAVES_M;
  const int wn = wave / WAVES_M;
  const int l16 = lane & 15;
  const int q = lane >> 4;
  const int lc = lane & 7, lr = lane >> 3;
  // image of a round: row r (32 rows) is 128 bytes = 8 chunks of 8 channels, chunk c at
  // (c ^ (r & 7)) * 16.  Block (i2, a, b) puts the lane's 4 channels into row b * 16 + l16,
  // chunk i2 * 4 + a * 2 + (q >> 1), half (q & 1): the chunk is (i2 * 4 + a * 2) XOR (q >> 1),
  // so its swizzled position is a per-lane constant XOR (i2 * 4 + a * 2)
  unsigned char* wr[4];
  {
    const int t = ((q >> 1) ^ (l16 & 7)) << 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) wr[c] = scratch + l16 * 128 + (q & 1) * 8 + (t ^ (c << 5));
  }
  const unsigned char* rd = scratch + lr * 128 + ((lc ^ lr) << 4);
  // Per-row and per-channel operands come from LDS tables, and a table read behind a write to
  // the image cannot be moved in front of it by the compiler (both are LDS): they are read where
  // no image write is pending -- the token scales of the wave's rows once, a 64-channel
  // column block's scales and biases at the top of the block, for both of its 32-row rounds.
  float sam[TM][2];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    sam[j][0] = sam[j][1] = 1.f;
    if constexpr (DT == 1) {
      sam[j][0] = sa_l[wm * TM * 32 + j * 32 + l16];
      sam[j][1] = sa_l[wm * TM * 32 + j * 32 + 16 + l16];
    }
  }
#pragma unroll
  for (int ip = 0; ip < TN / 2; ++ip) {
    // (with gate / residual in flight the registers do not reach: that copy reads them in place)
    constexpr bool TABLES_AHEAD = MODE != EPI_GATE && MODE != EPI_GENERAL;
    f32x4 sbv[4];
    u32x2 bb[4];
    auto read_tables = [&](int c) {  // c = i2 * 2 + a: 16 channels
      const int nl = wn * TN * 32 + ip * 64 + c * 16 + 4 * q;
      sbv[c] = f32x4{1.f, 1.f, 1.f, 1.f};
      if constexpr (DT == 1) sbv[c] = *reinterpret_cast<const f32x4*>(sb_l + nl);
      bb[c] = *reinterpret_cast<const u32x2*>(bias_l + nl);
    };
    if constexpr (TABLES_AHEAD) {
#pragma unroll
      for (int c = 0; c < 4; ++c) read_tables(c);
    }
    const int n8 = n0 + wn * TN * 32 + ip * 64 + lc * 8;
    const bool nok = n8 < p.N;
    const int n8c = min(n8, p.N - 8);  // N is a multiple of 8
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int mw = m0 + wm * TM * 32 + j * 32;
      // Output (and gate) rows of this 32-row block.  mw is wave-uniform, so the batch a row
      // falls into is found with ONE division per block when a batch has at least 32 rows (the
      // block then touches two batches at most); a per-row division otherwise.  Rows past the
      // end of the matrix take the last row's addresses (their loads are issued, their stores
      // are not): the block's first row is clamped like its other rows, so that every row lies
      // in the first row's batch or the next.
      const int mwc = min(mw, p.M - 1);
      int orow[4];  // output / residual row of store `it` (32-bit: rows, not offsets, stay live)
      const unsigned gr = gated ? (unsigned)p.gate_rows : 1u;
      const unsigned gb0 = gr >= 32 ? (unsigned)mwc / gr : 0u;
      auto gate_row = [&](int it) {  // element offset of the gate row of store `it`
        const unsigned mc = (unsigned)min(mw + it * 8 + lr, p.M - 1);
        const unsigned b = gr >= 32 ? gb0 + (mc >= (gb0 + 1) * gr ? 1u : 0u) : mc / gr;
        return (int64_t)b * p.ldg;
      };
      {
        const unsigned rpb = (unsigned)p.o_map.rows_per_batch;
        const unsigned ob0 = rpb >= 32 ? (unsigned)mwc / rpb : 0u;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int mc = min(mw + it * 8 + lr, p.M - 1);
          if (rpb == 0) {
            orow[it] = mc;
          } else if (rpb >= 32) {
            const unsigned b = ob0 + ((unsigned)mc >= (ob0 + 1) * rpb ? 1u : 0u);
            orow[it] = (int)(b * (unsigned)p.o_map.batch_rows + (unsigned)p.o_map.row_offset +
                             ((unsigned)mc - b * rpb));
          } else {
            orow[it] = (int)map_row32(mc, (int)rpb, p.o_map.batch_rows, p.o_map.row_offset);
          }
        }
      }
      // the residual of this round's four stores (from memory) is requested before the round's
      // arithmetic, the gate (a few rows, cached) before its trip through LDS
      u32x4 gv[4], rv[4];
      if (gated) {
#pragma unroll
        for (int it = 0; it < 4; ++it)
          rv[it] = *reinterpret_cast<const u32x4*>(p.residual + (int64_t)orow[it] * p.ldo + n8c);
      }
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) {
        const int i = 2 * ip + i2;
        // wave-uniform: the half-format column range is 32-aligned
        const int nblk = n0 + wn * TN * 32 + i * 32;
        const bool as_f16 = (ANY || MODE == EPI_F16) && nblk >= p.f16_begin && nblk < p.f16_end;
        float amax = 0.f;  // largest |value| of this 32-row x 32-channel block (half columns)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int c = i2 * 2 + a;
          if constexpr (!TABLES_AHEAD) read_tables(c);
          const f32x4 bsv =
              f32x4{__uint_as_float(bb[c][0] << 16), __uint_as_float(bb[c][0] & 0xffff0000u),
                    __uint_as_float(bb[c][1] << 16), __uint_as_float(bb[c][1] & 0xffff0000u)};
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            float y[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float v = acc[i * 2 + a][j * 2 + b][e];
              if constexpr (DT == 1) v = fmaf(v * sam[j][b], sbv[c][e], bsv[e]);
              else v += bsv[e];
              y[e] = v;
            }
            if (gelu) {
              const f32x2 g0 = gelu_tanh_f2(f32x2{rbf(y[0]), rbf(y[1])});
              const f32x2 g1 = gelu_tanh_f2(f32x2{rbf(y[2]), rbf(y[3])});
              y[0] = g0[0];
              y[1] = g0[1];
              y[2] = g1[0];
              y[3] = g1[1];
            }
            u32x2 pk;
            if (as_f16) {
              pk = u32x2{pack_bf2_as_f16(y[0], y[1]), pack_bf2_as_f16(y[2], y[3])};
#pragma unroll
              for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fabsf(y[e]));
            } else {
              pk = u32x2{pack_bf2(y[0], y[1]), pack_bf2(y[2], y[3])};
            }
            *reinterpret_cast<u32x2*>(wr[c] + b * 16 * 128) = pk;
          }
        }
        if (as_f16) note_f16_block(p.f16_counters, group_max<64>(amax), lane);
        if (gated && i2 == 0) {
#pragma unroll
          for (int it = 0; it < 4; ++it)
            gv[it] = *reinterpret_cast<const u32x4*>(p.gate + gate_row(it) + n8c);
        }
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        u32x4 o = *reinterpret_cast<const u32x4*>(rd + it * 8 * 128);
        if (gated) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float y0 = __uint_as_float(o[e] << 16), y1 = __uint_as_float(o[e] & 0xffff0000u);
            const float g0 = __uint_as_float(gv[it][e] << 16),
                        g1 = __uint_as_float(gv[it][e] & 0xffff0000u);
            const float r0 = __uint_as_float(rv[it][e] << 16),
                        r1 = __uint_as_float(rv[it][e] & 0xffff0000u);
            o[e] = pack_bf2(r0 + rbf(g0 * y0), r1 + rbf(g1 * y1));
          }
        }
        if (mw + it * 8 + lr < p.M && nok)
          __builtin_nontemporal_store(
              o, reinterpret_cast<u32x4*>(p.out + (int64_t)orow[it] * p.ldo + n8));
      }
    }
  }
}

template <int DT, int WAVES_M, int WAVES_N, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue16(const GemmParams& p, f32x4 (&acc)[TN * 2][TM * 2],
                                                unsigned char* scratch, const float* sb_l,
                                                const float* sa_l, const unsigned short* bias_l,
                                                int m0, int n0, int wave, int lane) {
  // wave-uniform (kernel arguments): one copy of the epilogue per combination a Flux step uses
  const bool gelu = p.act == MI355_ACT_GELU_TANH, halfcols = p.f16_end > p.f16_begin,
             gated = p.gate != nullptr;
#define MI355_EPI(MODE)                                                                          \
  gemm_epilogue16_mode<DT, WAVES_M, WAVES_N, TM, TN, MODE>(p, acc, scratch, sb_l, sa_l, bias_l, m0, \
                                                           n0, wave, lane)
  if (!gelu && !halfcols && !gated) MI355_EPI(EPI_PLAIN);
  else if (gelu && !halfcols && !gated) MI355_EPI(EPI_ACT);
  else if (halfcols && !gelu && !gated) MI355_EPI(EPI_F16);
  else if (gated && !gelu && !halfcols) MI355_EPI(EPI_GATE);
  else MI355_EPI(EPI_GENERAL);
#undef MI355_EPI
}

// accumulator tile of one wave: 32x32 blocks of 16 registers, or 16x16 blocks of 4
template <bool MF16, int TN, int TM>
struct AccTile {
  f32x16 v[TN][TM];
};
template <int TN, int TM>
struct AccTile<true, TN, TM> {
  f32x4 v[TN * 2][TM * 2];
};

// Persistent GEMM: the grid holds as many workgroups as fit on the chip at once and each
// walks the output tiles vb = blockIdx.x, blockIdx.x + gridDim.x, ...  The operand stages
// keep alternating across tile boundaries, so when a tile's K loop ends the stage it did NOT
// use last is idle: the first K tile of the NEXT output tile is sent there by LDS-DMA before
// the epilogue starts and lands underneath it, and the stage used last becomes the
// epilogue's scratch (one 4-KiB slice per wave) and table area.
// MF16: the 16x16 MFMA shapes instead of the 32x32 ones (plain per-token / bf16 form only.  The
// block-scaled form was built on v_mfma_scale_f32_16x16x128_f8f6f4 as well -- one MFMA from zero
// per scale block, four multiply-adds to retire it -- and ran 1-5 % slower than the 32x32 form
// below with units of two token blocks, 3-4 % with units of four: removed).
template <int DT /*0 bf16, 1 fp8*/, int WAVES_M, int WAVES_N, int TM, int TN, bool BLOCKSCALE,
          bool OUT16 = false, bool MF16 = false, bool PIPE = false>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, 2) void gemm_kernel(GemmParams p) {
  static_assert(!MF16 || (!BLOCKSCALE && !OUT16), "16x16 MFMAs: plain form only");
  static_assert(!PIPE || (MF16 && DT == 0), "pipelined K loop: bf16 on the 16x16 shape");
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int BM = WAVES_M * TM * 32;
  constexpr int BN = WAVES_N * TN * 32;
  constexpr int STAGE_BYTES = (BM + BN) * 128;
  constexpr int NPIECE = (BM + BN) / 8 / NW;  // 1-KiB LDS-DMA pieces per wave per K tile
  static_assert((BM + BN) / 8 % NW == 0, "pieces must divide over waves");
  constexpr int KSTEPS = DT == 0 ? 4 : 2;
  constexpr int SCRATCH_WAVE = 32 * 128;
  constexpr int TABLE_OFF = NW * SCRATCH_WAVE;
  static_assert(TABLE_OFF + BN * 4 + BM * 4 + BN * 2 <= STAGE_BYTES, "scratch + tables fit a stage");
  // block scales ride along with the operand stages: one K tile's BM activation scales and the
  // weight-block scales of the tile's BN / 128 column blocks (a 64-lane slot), behind the stages
  constexpr int SCALE_OFF = 2 * STAGE_BYTES;
  constexpr int SCALE_STAGE = (BM + 64) * 4;
  static_assert(!BLOCKSCALE || (NW > BM / 64 && BM % 64 == 0 && BN / 128 <= 64), "scale loader waves");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WAVES_M;
  const int wn = wave / WAVES_M;

  // ---- virtual block id -> output tile: XCD-contiguous chunks, 8-row groups -------
  const int tiles_m = (p.M + BM - 1) / BM;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  auto tile_of = [&](int vb, int& m0, int& n0) {
    // ids b and b+8 run on one XCD (and share its L2): give each XCD a contiguous range
    // of tile ids.  Bijective for any nwg.
    const int q = nwg >> 3, rr = nwg & 7, xcd = vb & 7;
    const int bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (vb >> 3);
    const int GROUP = p.group;
    const int group_sz = GROUP * tiles_n;
    const int gidx = bid / group_sz;
    const int first_m = gidx * GROUP;
    const int gm = min(tiles_m - first_m, GROUP);
    const int in_g = bid - gidx * group_sz;
    m0 = (first_m + in_g % gm) * BM;
    n0 = (in_g / gm) * BN;
  };

  // ---- per-lane LDS-DMA source pointers (advance 128 B per K tile) ---------
  const unsigned char* src[NPIECE];
  const float* sc_src = nullptr;  // block scales: this lane's scale of the next K tile to stage
  // (the lane id of the once-per-tile code is re-made opaque where it is used: what the compiler
  // would otherwise compute once in front of the tile loop and keep in registers across the K
  // loops does not fit beside the accumulators and fragments)
  auto opaque = [](int v) {
    asm volatile("" : "+v"(v));
    return v;
  };
  auto set_src = [&](int m0, int n0) {
    const int ln = opaque(lane);
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
      const int piece = wave + i * NW;
      const int row = piece * 8 + (ln >> 3);  // row inside [X tile | W tile]
      const int pc = ln & 7;
      const int lc = pc ^ ((row >> 1) & 7);
      if (piece < BM / 8) {
        int64_t gmrow = m0 + row;
        if (gmrow >= p.M) gmrow = p.M - 1;
        gmrow = map_row32((int)gmrow, (int)p.a_map.rows_per_batch, p.a_map.batch_rows,
                          p.a_map.row_offset);
        src[i] = p.a + gmrow * p.lda_b + lc * 16;
      } else {
        int64_t gnrow = n0 + (row - BM);
        if (gnrow >= p.N) gnrow = p.N - 1;
        src[i] = p.w + gnrow * p.ldw_b + lc * 16;
      }
    }
    if constexpr (BLOCKSCALE) {
      // waves 0 .. BM/64 - 1: lane l fetches the scale of token wave * 64 + l (group scales live
      // beside the activation rows: index by the mapped A row); wave BM/64: lane l the scale of
      // the tile's l-th 128-channel block (clamped: the extra lanes repeat the last one)
      if (wave < BM / 64) {
        const int m = min(m0 + wave * 64 + lane, p.M - 1);
        sc_src = p.sa + map_row32(m, (int)p.a_map.rows_per_batch, p.a_map.batch_rows,
                                  p.a_map.row_offset) * p.sa_sm;
      } else {
        const int nb = min((n0 >> 7) + min(lane, BN / 128 - 1), (p.N - 1) >> 7);
        sc_src = p.sb + nb * p.sb_sn;
      }
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
      const int piece = wave + i * NW;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)src[i],
                                       (lds_ptr_t)(smem + buf * STAGE_BYTES + piece * 1024), 16, 0,
                                       0);
      src[i] += 128;
    }
    if constexpr (BLOCKSCALE) {
      // 4 bytes per lane, straight into LDS like the operands: no register waits for them, and
      // the wait + barrier that opens a K tile covers them too
      if (wave <= BM / 64) {
        __builtin_amdgcn_global_load_lds(
            (gbl_ptr_t)sc_src, (lds_ptr_t)(smem + SCALE_OFF + buf * SCALE_STAGE + wave * 256), 4, 0, 0);
        sc_src += wave < BM / 64 ? p.sa_sk : p.sb_sk;
      }
    }
  };

  // per-tile epilogue operands, fetched into registers at tile start
  float sb_reg = 1.f, sa_reg = 1.f;
  unsigned short bias_reg = 0;
  auto load_tables = [&](int m0, int n0) {
    if (tid < BN) {
      const int n = n0 + tid;
      if constexpr (DT == 1 && !BLOCKSCALE) sb_reg = n < p.N ? p.sb[n] : 1.f;
      bias_reg = (p.bias && n < p.N) ? p.bias[n] : (unsigned short)0;
    }
    if constexpr (DT == 1 && !BLOCKSCALE) {
      if (tid < BM) {
        const int mc = min(m0 + tid, p.M - 1);
        // per-token scales live beside the activation rows: index by the mapped A row
        sa_reg = p.sa[map_row32(mc, (int)p.a_map.rows_per_batch, p.a_map.batch_rows,
                                p.a_map.row_offset)];
      }
    }
  };

  // ---- per-lane fragment read offsets --------------------------------------
  const int r = lane & 31;
  const int h = lane >> 5;
  const int swz = (r >> 1) & 7;
  int coff[KSTEPS];  // byte offset of the lane's first 16-B chunk for K step s
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) {
    const int c = DT == 0 ? (2 * s + h) : (4 * s + 2 * h);
    coff[s] = ((c ^ swz) << 4);
  }
  const int x_row_off = (wm * TM * 32 + r) * 128;
  const int w_row_off = BM * 128 + (wn * TN * 32 + r) * 128;
  const int unit_scale = 0x7f7f7f7f;  // E8M0 127 = 2^0 in every byte
  // 16x16 shapes: lane l reads row (l & 15) of a 16-row block and, of each K step, the 16-byte
  // chunk (l >> 4) (bf16: K step = 32 elements = 4 chunks; fp8: the whole 128-byte K tile is one
  // step and the lane takes chunks q and q + 4 -- any split of K over the lane groups works as
  // long as both operands use the same one, and this one keeps the reads conflict-free under
  // the row swizzle)
  const int l16 = lane & 15;
  const int q16 = lane >> 4;
  const int swz16 = (l16 >> 1) & 7;
  const int x_row_off16 = (wm * TM * 32 + l16) * 128;
  const int w_row_off16 = BM * 128 + (wn * TN * 32 + l16) * 128;
  const int coff16_0 = ((q16 ^ swz16) << 4);        // bf16 K step 0, fp8 low half
  const int coff16_1 = (((q16 + 4) ^ swz16) << 4);  // bf16 K step 1, fp8 high half

  int vb = blockIdx.x;
  int m0, n0;
  tile_of(vb, m0, n0);
  set_src(m0, n0);
  load_tables(m0, n0);
  int g = 0;  // K tiles consumed so far by this workgroup: stage parity runs across tiles
  stage(0);

  while (true) {
    AccTile<MF16, TN, TM> acct;
    auto& acc = acct.v;
    if constexpr (MF16) {
#pragma unroll
      for (int i = 0; i < TN * 2; ++i)
#pragma unroll
        for (int j = 0; j < TM * 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    }

    // software pipeline of the block-scaled K loop: two partial tiles, and the scale of the
    // unit still waiting to be added (the last unit of the previous K tile)
    static_assert(!BLOCKSCALE || (TN * TM) % 2 == 0, "the last unit of a K tile uses part[1]");
    // a wave's channels (TN * 32 <= 128, aligned) lie inside one 128-channel scale block
    static_assert(!BLOCKSCALE || (TN * 32 <= 128 && 128 % (TN * 32) == 0),
                  "one weight scale per wave and K tile");
    f32x16 part[2];
    float pend_f = 0.f;
    if constexpr (BLOCKSCALE) {
#pragma unroll
      for (int e = 0; e < 16; ++e) part[1][e] = 0.f;
    }

    if constexpr (PIPE) {
      // Software-pipelined K loop (bf16 only.  The fp8 forms of it that fit 256 registers beside
      // the accumulators -- one activation set, re-read after a tile's last unit -- ran 1-7 %
      // SLOWER than the plain loop.  With asm reads an fp8 operand is two 16-byte reads into ONE
      // 8-register tuple: as two asm outputs the copies derail the schedule (136 spills); with
      // each fragment in a FIXED tuple (the output constraint names the tuple, the instruction
      // text its halves) and the MFMAs as volatile asm as well, the loop comes out exactly as
      // written, same bits, and still 1-8 % slower at one unit of prefetch (18 spills) and
      // 10-20 % slower at two (61 spills).  All removed.)  A K tile is walked in units of two weight
      // fragments = 8 MFMAs; the fragments of unit u + 1 are requested from LDS
      // before the MFMAs of unit u are issued, and the wait + barrier that opens the NEXT K tile
      // sits inside the LAST unit of this one: by then every wave has all of this tile's
      // fragments in registers, so the stage can be refilled while the last MFMAs run.
      {
        const int cur0 = g & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (p.ktiles > 1) stage(cur0 ^ 1);
      }
      {
        // The fragment reads are inline asm with hand-counted waits: left to the compiler, every
        // second unit waited with lgkmcnt(0) right after requesting the NEXT unit's fragments,
        // i.e. for the reads it had just issued.  LDS operations return in order and no other
        // LDS or scalar-memory operation is issued inside the K loop, so "all but the n youngest"
        // is exact.  Every wait is followed by a sched_barrier: the compiler would otherwise move
        // register-only MFMAs above it.
        constexpr int NIG = TN;      // units per K step (two 16-row weight fragments each)
        constexpr int NU = 2 * NIG;  // units per K tile (two K steps of 32)
        bf16x8 afb[2][2], bfb[2][TM * 2];
        const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
#define MI355_LDS_READ(dst, addr, imm) \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm) : "memory")
#define MI355_LDS_WAIT(n)                                    \
  do {                                                       \
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory"); \
    __builtin_amdgcn_sched_barrier(0);                       \
  } while (0)
        unsigned xa0 = lds0 + (g & 1) * STAGE_BYTES + x_row_off16 + coff16_0;
        unsigned xa1 = lds0 + (g & 1) * STAGE_BYTES + x_row_off16 + coff16_1;
        unsigned wa0 = lds0 + (g & 1) * STAGE_BYTES + w_row_off16 + coff16_0;
        unsigned wa1 = lds0 + (g & 1) * STAGE_BYTES + w_row_off16 + coff16_1;
#pragma unroll
        for (int j = 0; j < TM * 2; ++j) MI355_LDS_READ(bfb[0][j], xa0, j * 16 * 128);
#pragma unroll
        for (int a = 0; a < 2; ++a) MI355_LDS_READ(afb[0][a], wa0, a * 16 * 128);
        for (int kt = 0; kt < p.ktiles; ++kt, ++g) {
          const bool turn = kt + 1 < p.ktiles;  // wave-uniform
#pragma unroll
          for (int u = 0; u < NU; ++u) {
            const int ks = u / NIG, ig = u % NIG;
            if (u + 1 < NU) {
              const int ks1 = (u + 1) / NIG, ig1 = (u + 1) % NIG;
              if (ig1 == 0) {
#pragma unroll
                for (int j = 0; j < TM * 2; ++j) MI355_LDS_READ(bfb[ks1][j], ks1 ? xa1 : xa0, j * 16 * 128);
              }
#pragma unroll
              for (int a = 0; a < 2; ++a)
                MI355_LDS_READ(afb[(u + 1) & 1][a], ks1 ? wa1 : wa0, (2 * ig1 + a) * 16 * 128);
              if (ig1 == 0) MI355_LDS_WAIT(TM * 2 + 2);
              else MI355_LDS_WAIT(2);
            } else if (turn) {
              // every fragment of this K tile has to be in registers before the stage is refilled
              asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
              __syncthreads();
              if (kt + 2 < p.ktiles) stage(g & 1);
              {
                // the other stage (a stage is not a power of two for every tile shape)
                const unsigned nb = lds0 + ((g + 1) & 1) * STAGE_BYTES;
                xa0 = nb + x_row_off16 + coff16_0;
                xa1 = nb + x_row_off16 + coff16_1;
                wa0 = nb + w_row_off16 + coff16_0;
                wa1 = nb + w_row_off16 + coff16_1;
              }
#pragma unroll
              for (int j = 0; j < TM * 2; ++j) MI355_LDS_READ(bfb[0][j], xa0, j * 16 * 128);
#pragma unroll
              for (int a = 0; a < 2; ++a) MI355_LDS_READ(afb[0][a], wa0, a * 16 * 128);
              __builtin_amdgcn_sched_barrier(0);
            } else {
              MI355_LDS_WAIT(0);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
              for (int j = 0; j < TM * 2; ++j)
                acc[2 * ig + a][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    afb[u & 1][a], bfb[ks][j], acc[2 * ig + a][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
#undef MI355_LDS_READ
#undef MI355_LDS_WAIT
      }
    } else
    for (int kt = 0; kt < p.ktiles; ++kt, ++g) {
      const int cur = g & 1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (kt + 1 < p.ktiles) stage(cur ^ 1);
      const unsigned char* sbase = smem + cur * STAGE_BYTES;

      if constexpr (MF16) {
        const unsigned char* xb = sbase + x_row_off16;
        const unsigned char* wb = sbase + w_row_off16;
        if constexpr (DT == 0) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const int co = s ? coff16_1 : coff16_0;
            bf16x8 af[TN * 2], bfr[TM * 2];
#pragma unroll
            for (int j = 0; j < TM * 2; ++j)
              bfr[j] = *reinterpret_cast<const bf16x8*>(xb + j * 16 * 128 + co);
#pragma unroll
            for (int i = 0; i < TN * 2; ++i)
              af[i] = *reinterpret_cast<const bf16x8*>(wb + i * 16 * 128 + co);
#pragma unroll
            for (int i = 0; i < TN * 2; ++i)
#pragma unroll
              for (int j = 0; j < TM * 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
          }
        } else {
          i32x8 bfr[TM * 2];
#pragma unroll
          for (int j = 0; j < TM * 2; ++j) {
            const unsigned char* pb = xb + j * 16 * 128;
            const i32x4 lo = *reinterpret_cast<const i32x4*>(pb + coff16_0);
            const i32x4 hi = *reinterpret_cast<const i32x4*>(pb + coff16_1);
            bfr[j] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          }
          // the weight fragments in groups of four (a 256-register budget at 2 waves per SIMD)
#pragma unroll
          for (int ig = 0; ig < TN * 2; ig += 4) {
            if constexpr (TM * TN >= 8) __builtin_amdgcn_sched_barrier(0);
            i32x8 af[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const unsigned char* pa = wb + (ig + i) * 16 * 128;
              const i32x4 lo = *reinterpret_cast<const i32x4*>(pa + coff16_0);
              const i32x4 hi = *reinterpret_cast<const i32x4*>(pa + coff16_1);
              af[i] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int j = 0; j < TM * 2; ++j)
                acc[ig + i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                    af[i], bfr[j], acc[ig + i][j], 0, 0, 0, unit_scale, 0, unit_scale);
          }
        }
      } else if constexpr (BLOCKSCALE) {
        // One K tile is one 128-wide scale block.  The K tile is walked in TN*TM units of one
        // 32x32 output block each: a unit's MFMAs (a chain over the K steps, started from
        // zero) go into one of two partial tiles, and while they run the PREVIOUS unit's
        // partial tile is scaled into the accumulators on the VALU -- the matrix pipe never
        // waits for the scaling, and only two partial tiles exist (a second accumulator set
        // would not fit 2 waves per SIMD).  The last unit of a K tile is retired under the
        // first unit of the next one.
        // The K tile's scales were staged into LDS with its operands.
        const float* sc_l = reinterpret_cast<const float*>(smem + SCALE_OFF + cur * SCALE_STAGE);
        const float sbw = sc_l[BM + ((wn * TN * 32) >> 7)];
        float fa[TM];
#pragma unroll
        for (int j = 0; j < TM; ++j) fa[j] = sc_l[wm * TM * 32 + j * 32 + r] * sbw;
        i32x8 xf[KSTEPS][TM];
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
          for (int j = 0; j < TM; ++j) {
            const unsigned char* pb = sbase + x_row_off + j * 32 * 128;
            i32x4 lo = *reinterpret_cast<const i32x4*>(pb + coff[s]);
            i32x4 hi = *reinterpret_cast<const i32x4*>(pb + (coff[s] ^ 16));
            xf[s][j] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          }
        i32x8 wf[KSTEPS];
        auto load_w = [&](int i) {
#pragma unroll
          for (int s = 0; s < KSTEPS; ++s) {
            const unsigned char* pa = sbase + w_row_off + i * 32 * 128;
            i32x4 lo = *reinterpret_cast<const i32x4*>(pa + coff[s]);
            i32x4 hi = *reinterpret_cast<const i32x4*>(pa + (coff[s] ^ 16));
            wf[s] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          }
        };
        load_w(0);
#pragma unroll
        for (int u = 0; u < TN * TM; ++u) {
          const int i = u / TM, j = u % TM;
          {
            f32x16 zero;
#pragma unroll
            for (int e = 0; e < 16; ++e) zero[e] = 0.f;
            part[u & 1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                wf[0], xf[0][j], zero, 0, 0, 0, unit_scale, 0, unit_scale);
          }
#pragma unroll
          for (int s = 1; s < KSTEPS; ++s)
            part[u & 1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                wf[s], xf[s][j], part[u & 1], 0, 0, 0, unit_scale, 0, unit_scale);
          __builtin_amdgcn_sched_barrier(0);
          if (j == TM - 1 && i + 1 < TN) load_w(i + 1);  // lands under the queued MFMAs
          __builtin_amdgcn_sched_barrier(0);
          if (u == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[TN - 1][TM - 1][e] += part[1][e] * pend_f;
          } else {
#pragma unroll
            for (int e = 0; e < 16; ++e)
              acc[(u - 1) / TM][(u - 1) % TM][e] += part[(u - 1) & 1][e] * fa[(u - 1) % TM];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        pend_f = fa[TM - 1];
      } else {
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
          // keep one K step's fragments live at a time (256-VGPR budget at 2 waves/SIMD)
          if constexpr (DT == 1 && TM * TN >= 8) __builtin_amdgcn_sched_barrier(0);
          if constexpr (DT == 0) {
            bf16x8 af[TN], bfr[TM];
#pragma unroll
            for (int i = 0; i < TN; ++i)
              af[i] = *reinterpret_cast<const bf16x8*>(sbase + w_row_off + i * 32 * 128 + coff[s]);
#pragma unroll
            for (int j = 0; j < TM; ++j)
              bfr[j] = *reinterpret_cast<const bf16x8*>(sbase + x_row_off + j * 32 * 128 + coff[s]);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
              for (int j = 0; j < TM; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
          } else {
            i32x8 af[TN], bfr[TM];
#pragma unroll
            for (int i = 0; i < TN; ++i) {
              const unsigned char* pa = sbase + w_row_off + i * 32 * 128;
              i32x4 lo = *reinterpret_cast<const i32x4*>(pa + coff[s]);
              i32x4 hi = *reinterpret_cast<const i32x4*>(pa + (coff[s] ^ 16));
              af[i] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int j = 0; j < TM; ++j) {
              const unsigned char* pb = sbase + x_row_off + j * 32 * 128;
              i32x4 lo = *reinterpret_cast<const i32x4*>(pb + coff[s]);
              i32x4 hi = *reinterpret_cast<const i32x4*>(pb + (coff[s] ^ 16));
              bfr[j] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
              for (int j = 0; j < TM; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                    af[i], bfr[j], acc[i][j], 0, 0, 0, unit_scale, 0, unit_scale);
          }
        }
      }
    }

    if constexpr (BLOCKSCALE) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[TN - 1][TM - 1][e] += part[1][e] * pend_f;
    }

    // ---- tile done: park the tables, prefetch the next tile, run the epilogue ----------
    const int lastb = (g - 1) & 1;
    __syncthreads();  // every wave is done reading operand stages
    unsigned char* area = smem + lastb * STAGE_BYTES;
    float* sb_l = reinterpret_cast<float*>(area + TABLE_OFF);
    float* sa_l = sb_l + BN;
    unsigned short* bias_l = reinterpret_cast<unsigned short*>(sa_l + BM);
    if (tid < BN) {
      sb_l[tid] = sb_reg;
      bias_l[tid] = bias_reg;
    }
    if (tid < BM) sa_l[tid] = sa_reg;
    const int em0 = m0, en0 = n0;
    const int nvb = vb + (int)gridDim.x;
    const bool more = nvb < nwg;
    if (more) {
      tile_of(nvb, m0, n0);
      set_src(m0, n0);
      stage(lastb ^ 1);  // first K tile of the next output tile, in flight under the epilogue
      load_tables(m0, n0);
    }
    __syncthreads();  // tables visible
    if constexpr (MF16)
      gemm_epilogue16<DT, WAVES_M, WAVES_N, TM, TN>(p, acc, area + wave * SCRATCH_WAVE, sb_l, sa_l,
                                                    bias_l, em0, en0, wave, opaque(lane));
    else
      gemm_epilogue<DT, WAVES_M, WAVES_N, TM, TN, BLOCKSCALE, OUT16>(
          p, acc, area + wave * SCRATCH_WAVE, sb_l, sa_l, bias_l, em0, en0, wave, opaque(lane));
    if (!more) break;
    vb = nvb;
  }
}

// ---- fp8 W8A8 GEMM, 256 x 256 tiles, operands through a five-slot LDS ring -------------
// What bounds the 256 x 256 fp8 tile is not its MFMAs (2048 clocks per 128-byte K tile and
// SIMD) but the L2 -> LDS fill: 64 KiB per K tile and CU arrive at 25-30 B/clk, ~2200 clocks,
// and in a two-stage scheme (everything requested behind one barrier, everything awaited in
// front of the next) the fill pipe also runs EMPTY around every barrier -- measured with the
// MFMAs removed, a K tile still took 2400-3000 clocks; with the refill removed, 2500.
// Here an operand tile of one K tile (256 rows x 128 B = 32 KiB) is an item of a stream
//     A_0 W_0 A_1 W_1 A_2 ...      item t lives in slot t mod 5 (5 x 32 KiB = all of the LDS)
// and behind the barrier that opens K tile g the wave requests W_{g+1} and A_{g+2} (into the
// slots A_{g-1} and W_{g-1} have just left) and waits, in front of the next barrier, with
// vmcnt(4): for everything but its four youngest pieces, i.e. A_{g+2} stays in flight ACROSS
// the barrier and the fill pipe never drains.  The stream runs on across output tiles (the
// activation pointers move to the workgroup's next output tile two K tiles before the end, the
// weight pointers one), so there is no start-up per output tile either; when no tile follows,
// the stream re-reads the current tile's first K tiles (valid addresses, unused results).
// The epilogue borrows the two slots the last K tile has just left (scratch, tables); the
// barrier that follows it refills them.  Wave tiles, fragment layout, MFMA order and epilogue
// are those of gemm_kernel<1, 4, 2, 2, 4, ..., MF16>: same bits.
// TAIL: the launch has a partial last round that is run as half tiles (below); without it no
// tile is ever a half and the kernel carries none of that state.
template <bool TAIL>
__global__ __launch_bounds__(512, 2) void gemm_ring_kernel(GemmParams p) {
  constexpr int NW = 8, WAVES_M = 4, TM = 2, TN = 4, BM = 256, BN = 256;
  constexpr int SLOT = 256 * 128;
  constexpr int NSLOT = 5;
  constexpr int NP = 4;  // 1-KiB LDS-DMA pieces per wave and item
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WAVES_M, wn = wave / WAVES_M;

  const int tiles_m = (p.M + BM - 1) / BM;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  // The walk of gemm_kernel over the tiles of the full rounds; the tiles of a partial last round
  // (p.split_from .. nwg - 1, when they leave at least half of the workgroups without one) are
  // cut into a left and a right half of 128 channels, so that every workgroup has a last piece of
  // work and that piece is short: a wave's 64 x 64 outputs then, half the MFMAs and three
  // quarters of the staged bytes per K tile.
  const int split_from = TAIL ? p.split_from : nwg;
  const int nhalf = 2 * (nwg - split_from);
  const int nvirt = split_from + nhalf;
  auto tile_of = [&](int vb, int& m0, int& n0, bool& half) {
    int tile;
    int hsel = 0;
    half = TAIL && vb >= split_from;
    if (!half) {
      const int q = split_from >> 3, rr = split_from & 7, xcd = vb & 7;
      tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (vb >> 3);
    } else {
      const int hv = vb - split_from;
      const int q = nhalf >> 3, rr = nhalf & 7, xcd = hv & 7;
      const int h = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (hv >> 3);
      tile = split_from + (h >> 1);
      hsel = h & 1;
    }
    const int group_sz = p.group * tiles_n;
    const int gidx = tile / group_sz;
    const int first_m = gidx * p.group;
    const int gm = min(tiles_m - first_m, p.group);
    const int in_g = tile - gidx * group_sz;
    m0 = (first_m + in_g % gm) * BM;
    n0 = (in_g / gm) * BN + hsel * (BN / 2);
  };

  // per-lane sources of the two streams: piece = wave + 8 i holds rows piece * 8 .. + 7 of the
  // operand tile, 16-byte chunks swizzled by the row as in gemm_kernel
  // (32-bit byte offsets from the operand bases: the launcher sends operands of 4 GiB and more to
  // the two-stage kernel; eight vector registers less than per-lane pointers)
  unsigned int srca[NP], srcw[NP];
  // (the lane id of the once-per-tile code is re-made opaque where it is used: what the compiler
  // would otherwise compute once in front of the tile loop and keep in registers across the K
  // loops does not fit beside 128 accumulators and 64 fragment registers)
  auto opaque = [](int v) {
    asm volatile("" : "+v"(v));
    return v;
  };
  auto set_src_a = [&](int m0) {
    const int ln = opaque(lane);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int row = (wave + i * NW) * 8 + (ln >> 3);
      const int lc = (ln & 7) ^ ((row >> 1) & 7);
      const int64_t gm = map_row32(min(m0 + row, p.M - 1), (int)p.a_map.rows_per_batch,
                                   p.a_map.batch_rows, p.a_map.row_offset);
      srca[i] = (unsigned int)(gm * p.lda_b + lc * 16);
    }
  };
  auto set_src_w = [&](int n0) {
    const int ln = opaque(lane);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int row = (wave + i * NW) * 8 + (ln >> 3);
      const int lc = (ln & 7) ^ ((row >> 1) & 7);
      srcw[i] = (unsigned int)((int64_t)min(n0 + row, p.N - 1) * p.ldw_b + lc * 16);
    }
  };
  auto piece_a = [&](int slot, int i) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(p.a + srca[i]),
                                     (lds_ptr_t)(smem + slot * SLOT + (wave + i * NW) * 1024), 16, 0, 0);
    srca[i] += 128;
  };
  auto piece_w = [&](int slot, int i) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(p.w + srcw[i]),
                                     (lds_ptr_t)(smem + slot * SLOT + (wave + i * NW) * 1024), 16, 0, 0);
    srcw[i] += 128;
  };

  float sb_reg = 1.f, sa_reg = 1.f;
  unsigned short bias_reg = 0;
  auto load_tables = [&](int m0, int n0) {
    if (tid < BN) {
      const int n = n0 + tid;
      sb_reg = n < p.N ? p.sb[n] : 1.f;
      bias_reg = (p.bias && n < p.N) ? p.bias[n] : (unsigned short)0;
    }
    if (tid < BM) {
      const int mc = min(m0 + tid, p.M - 1);
      sa_reg = p.sa[map_row32(mc, (int)p.a_map.rows_per_batch, p.a_map.batch_rows,
                              p.a_map.row_offset)];
    }
  };

  const int l16 = lane & 15, q16 = lane >> 4, swz16 = (l16 >> 1) & 7;
  const int x_off = (wm * TM * 32 + l16) * 128;
  const int w_off_full = (wn * TN * 32 + l16) * 128, w_off_half = (wn * TN * 16 + l16) * 128;
  const int c0 = (q16 ^ swz16) << 4, c1 = ((q16 + 4) ^ swz16) << 4;
  const int unit_scale = 0x7f7f7f7f;

  auto next_slot = [](int s, int by) {
    s += by;
    return s >= NSLOT ? s - NSLOT : s;
  };

  int vb = blockIdx.x;
  int m0, n0;
  bool half;
  tile_of(vb, m0, n0, half);
  int nm0 = m0, nn0 = n0, nvb = vb;
  bool more = false, nhalf_t = false;
  // start of the stream: A_0, W_0, A_1 (at least two K tiles per output tile: launch precondition)
  set_src_a(m0);
  set_src_w(n0);
  load_tables(m0, n0);
#pragma unroll
  for (int i = 0; i < NP; ++i) piece_a(0, i);
#pragma unroll
  for (int i = 0; i < NP; ++i)
    if (i < 2 || !(TAIL && half)) piece_w(1, i);
#pragma unroll
  for (int i = 0; i < NP; ++i) piece_a(2, i);
  int sl = 0;  // slot of the current K tile's activation item (its weight item: the next slot)

  while (true) {
    f32x4 acc[TN * 2][TM * 2];
#pragma unroll
    for (int i = 0; i < TN * 2; ++i)
#pragma unroll
      for (int j = 0; j < TM * 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // one K loop per tile width (a half tile is always a workgroup's last tile): no branch inside
    // a K tile
    auto k_loop = [&](auto halfc) {
      constexpr bool HALF = decltype(halfc)::value;
    for (int kt = 0; kt < p.ktiles; ++kt) {
      // this K tile's two items have landed (everything but the youngest item, A of the next K
      // tile; after an epilogue its stores are in the count as well: wait for all of it once)
      // (s_barrier itself, not __syncthreads(): the fence of the latter makes the compiler drain
      // vmcnt to zero in front of every barrier, which is exactly what the ring is there to avoid;
      // the waits a barrier over LDS needs are spelled out)
      if (kt == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // the streams move on to the next output tile: activations at K tile ktiles - 2 (they run
      // two K tiles ahead), weights at ktiles - 1
      if (kt + 2 == p.ktiles) {
        nvb = vb + (int)gridDim.x;
        more = nvb < nvirt;
        nm0 = m0;
        nn0 = n0;
        nhalf_t = false;  // the unused tail of the stream is staged at full width
        if (more) tile_of(nvb, nm0, nn0, nhalf_t);
        if (!TAIL) nhalf_t = false;
        set_src_a(nm0);
      }
      if (kt + 1 == p.ktiles) set_src_w(nn0);
      // the weight item requested here belongs to this tile, or (last K tile) to the next one
      const bool w_item_half = TAIL && (kt + 1 == p.ktiles ? nhalf_t : HALF);
      const int s_w1 = next_slot(sl, 3), s_a2 = next_slot(sl, 4);
      const unsigned char* xb = smem + sl * SLOT + x_off;
      const unsigned char* wb = smem + next_slot(sl, 1) * SLOT + (HALF ? w_off_half : w_off_full);
      piece_w(s_w1, 0);
      piece_w(s_w1, 1);
      __builtin_amdgcn_sched_barrier(0);
      i32x8 bfr[TM * 2];
#pragma unroll
      for (int j = 0; j < TM * 2; ++j) {
        const unsigned char* pb = xb + j * 16 * 128;
        const i32x4 lo = *reinterpret_cast<const i32x4*>(pb + c0);
        const i32x4 hi = *reinterpret_cast<const i32x4*>(pb + c1);
        bfr[j] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int ig = 0; ig < (HALF ? TN : TN * 2); ig += 4) {  // a half tile: four weight fragments per wave
        __builtin_amdgcn_sched_barrier(0);
        i32x8 af[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const unsigned char* pa = wb + (ig + i) * 16 * 128;
          const i32x4 lo = *reinterpret_cast<const i32x4*>(pa + c0);
          const i32x4 hi = *reinterpret_cast<const i32x4*>(pa + c1);
          af[i] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // the rest of the refill between the MFMA blocks: weights first (due at the next
          // barrier), then the activation item that stays in flight across it
          const int blk = ig + i;
          if (blk == 1) {
            if (!w_item_half) {
              piece_w(s_w1, 2);
              piece_w(s_w1, 3);
            }
          } else if (blk == 2 || blk == 3) {
            piece_a(s_a2, 2 * (blk - 2));
            piece_a(s_a2, 2 * (blk - 2) + 1);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < TM * 2; ++j)
            acc[ig + i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                af[i], bfr[j], acc[ig + i][j], 0, 0, 0, unit_scale, 0, unit_scale);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      sl = next_slot(sl, 2);
    }
    };
    if (TAIL && half) k_loop(std::true_type{});
    else k_loop(std::false_type{});

    // the two slots of the last K tile are free once every wave is here: scratch in the
    // activation slot, tables in the weight slot
    const int s_scr = next_slot(sl, NSLOT - 2), s_tab = next_slot(sl, NSLOT - 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    float* sb_l = reinterpret_cast<float*>(smem + s_tab * SLOT);
    float* sa_l = sb_l + BN;
    unsigned short* bias_l = reinterpret_cast<unsigned short*>(sa_l + BM);
    if (tid < BN) {
      sb_l[tid] = sb_reg;
      bias_l[tid] = bias_reg;
    }
    if (tid < BM) sa_l[tid] = sa_reg;
    const int em0 = m0, en0 = n0;
    if (more) load_tables(nm0, nn0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // tables visible
    asm volatile("" ::: "memory");
    const int eln = opaque(lane);
    if (TAIL && half)
      gemm_epilogue16<1, 4, 2, 2, 2>(p, reinterpret_cast<f32x4(&)[TN][TM * 2]>(acc),
                                     smem + s_scr * SLOT + wave * 4096, sb_l, sa_l, bias_l, em0, en0, wave,
                                     eln);
    else
      gemm_epilogue16<1, 4, 2, 2, 4>(p, acc, smem + s_scr * SLOT + wave * 4096, sb_l, sa_l, bias_l, em0,
                                     en0, wave, eln);
    if (!more) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the unused tail of the stream
      break;
    }
    vb = nvb;
    m0 = nm0;
    n0 = nn0;
    half = TAIL && nhalf_t;
  }
}

// ---- small-M (<= 16 rows) bf16 linear: pure weight streaming ------------------
// out[m][n] = act( sum_k x[m][k] w[n][k] + bias[n] ).  One wavefront per output
// channel pair; lanes split K in 16-byte chunks, fp32 accumulate, shuffle reduce.
// Weights are read exactly once from HBM; x (a few KB) stays in L1/L2.
template <int MMAX, int SKINNY_CH>
__global__ __launch_bounds__(256) void skinny_linear_kernel(
    const unsigned short* __restrict__ x, const unsigned short* __restrict__ w,
    const unsigned short* __restrict__ bias, unsigned short* __restrict__ out, int M, int N, int K,
    int64_t ldx, int64_t ldw, int64_t ldo, int act) {
  // A wave owns SKINNY_CH consecutive channels: every activation vector it fetches (from
  // cache) is used against SKINNY_CH weight vectors (from HBM), so the cached re-reads of x
  // stay below the weight stream instead of being MMAX times it.  Each (m, n) sum keeps the
  // order it had with one channel per wave: lane-strided over K, then the shuffle tree.
  constexpr int CH = SKINNY_CH;
  const int lane = threadIdx.x & 63;
  const int n0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * CH;
  if (n0 >= N) return;
  const u32x4* wr[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c)
    wr[c] = reinterpret_cast<const u32x4*>(w + (int64_t)min(n0 + c, N - 1) * ldw);
  float acc[CH][MMAX];
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int m = 0; m < MMAX; ++m) acc[c][m] = 0.f;
  const int nvec = K >> 3;
  for (int idx = lane; idx < nvec; idx += 64) {
    u32x4 wv[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) wv[c] = wr[c][idx];
#pragma unroll
    for (int m = 0; m < MMAX; ++m) {
      if (m < M) {
        const u32x4 xv = *reinterpret_cast<const u32x4*>(x + (int64_t)m * ldx + (int64_t)idx * 8);
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[c][m] = fmaf(__uint_as_float(xv[j] << 16), __uint_as_float(wv[c][j] << 16),
                             acc[c][m]);
            acc[c][m] = fmaf(__uint_as_float(xv[j] & 0xffff0000u),
                             __uint_as_float(wv[c][j] & 0xffff0000u), acc[c][m]);
          }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int m = 0; m < MMAX; ++m) acc[c][m] = group_sum<64>(acc[c][m]);
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int n = n0 + c;
      if (n < N) {
        const float b = bias ? bf2f(bias[n]) : 0.f;
#pragma unroll
        for (int m = 0; m < MMAX; ++m) {
          if (m < M) {
            float v = acc[c][m] + b;
            if (act == MI355_ACT_GELU_TANH) v = gelu_tanh_f(rbf(v));
            out[(int64_t)m * ldo + n] = f2bf(v);
          }
        }
      }
    }
  }
}

// workgroups of one kernel that are resident at once: CUs x (workgroups per CU by LDS)
int resident_workgroups(int lds_bytes) {
  static std::atomic<int> cus_of[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::atomic<int>& slot = cus_of[dev & 63];
  int cus = slot.load(std::memory_order_relaxed);
  if (cus == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    cus = n;
    slot.store(n, std::memory_order_relaxed);
  }
  const int per_cu = lds_bytes > 80 * 1024 ? 1 : 2;
  return cus * per_cu;
}

template <int DT, int WAVES_M, int WAVES_N, int TM, int TN, bool BS, bool OUT16 = false,
          bool MF16 = false, bool PIPE = false>
int launch_gemm(const GemmParams& p, hipStream_t st) {
  constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
  constexpr int LDS = 2 * (BM + BN) * 128 + (BS ? 2 * (BM + 64) * 4 : 0);
  // the LDS limit is a per-device function attribute: remember which devices have it
  // callers may launch from several host threads: the per-device flags are atomic (a lost
  // update would only repeat the attribute call)
  static std::atomic<unsigned long long> attr_devices{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long dev_bit = 1ull << (dev & 63);
  auto kfn = gemm_kernel<DT, WAVES_M, WAVES_N, TM, TN, BS, OUT16, MF16, PIPE>;
  if (!(attr_devices.load(std::memory_order_acquire) & dev_bit)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      mi355_set_error("gemm: cannot reserve %d B of LDS: %s", LDS, hipGetErrorString(e));
      return MI355_EHIP;
    }
    attr_devices.fetch_or(dev_bit, std::memory_order_release);
  }
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  GemmParams q = p;
  // M tiles per XCD chunk of the tile walk: 8 for wide outputs; 2 for narrow ones (N <= 4096,
  // e.g. proj_out with K = 15360: 836 -> 793 us; measured with tools/gemm_probe.py narrow)
  q.group = tiles_n <= 16 ? 2 : 8;
  dim3 grid((unsigned)min(nwg, resident_workgroups(LDS))), block(WAVES_M * WAVES_N * 64);
  hipLaunchKernelGGL(kfn, grid, block, LDS, st, q);
  MI355_CHECK_LAUNCH("gemm");
  return MI355_OK;
}

int launch_gemm_ring(const GemmParams& p, hipStream_t st) {
  constexpr int LDS = 5 * 256 * 128;  // all of a CU's LDS
  static std::atomic<unsigned long long> attr_devices{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long dev_bit = 1ull << (dev & 63);
  if (!(attr_devices.load(std::memory_order_acquire) & dev_bit)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ring_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ring_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      mi355_set_error("gemm: cannot reserve %d B of LDS: %s", LDS, hipGetErrorString(e));
      return MI355_EHIP;
    }
    attr_devices.fetch_or(dev_bit, std::memory_order_release);
  }
  const int tiles_m = (p.M + 255) / 256, tiles_n = (p.N + 255) / 256;
  GemmParams q = p;
  q.group = tiles_n <= 16 ? 2 : 8;
  const int nwg = tiles_m * tiles_n, wgs = min(nwg, resident_workgroups(LDS));
  // a partial last round that leaves at least half of the workgroups idle is run as half tiles
  const int tail = nwg % wgs;
  q.split_from = (tail > 0 && 2 * tail <= wgs) ? nwg - tail : nwg;
  dim3 grid((unsigned)wgs), block(512);
  if (q.split_from < nwg) hipLaunchKernelGGL(gemm_ring_kernel<true>, grid, block, LDS, st, q);
  else hipLaunchKernelGGL(gemm_ring_kernel<false>, grid, block, LDS, st, q);
  MI355_CHECK_LAUNCH("gemm_ring");
  return MI355_OK;
}

// 256x256 tiles (one workgroup per CU) or 128x128 (two per CU)?  Small ones when the big ones
// would not cover three quarters of the CUs; otherwise whichever fills its rounds better, the small tile counted at its measured relative speed (0.8 of the big one for fp8, 0.7
// for bf16): 2048 x 9216 fp8 is 288 big tiles = 1.125 rounds (93 us) but 2.25 rounds of small ones
// (72 us).  sglang_diffusion_amd/models/flux_engine.py labels its launches with the same rule.
bool use_big_tile(int64_t M, int64_t N, bool fp8) {
  const int cus = resident_workgroups(128 * 1024);
  const int64_t tb = ((M + 255) / 256) * ((N + 255) / 256);
  const int64_t ts = ((M + 127) / 128) * ((N + 127) / 128);
  if (tb < cus * 3 / 4) return false;  // too few big tiles to cover the chip: more, smaller ones
  const double eff_b = (double)tb / (double)(((tb + cus - 1) / cus) * cus);
  const double eff_s = (double)ts / (double)(((ts + 2 * cus - 1) / (2 * cus)) * (2 * cus));
  return eff_b >= eff_s * (fp8 ? 0.8 : 0.7);
}

// 128x64 tiles instead of 128x128: when even the small tiles leave half of the CUs without one
// (M = 512 text rows x N = 3072 is 96 tiles of 128x128 on 256 CUs, each walking the whole K)
bool use_narrow_tile(int64_t M, int64_t N) {
  const int cus = resident_workgroups(128 * 1024);
  return ((M + 127) / 128) * ((N + 127) / 128) * 2 <= cus;
}

int check_common(const mi355_linear_args* a, const char* who) {
  MI355_REQUIRE(a->M == 0 || (a->a && a->w && a->out), "%s: null pointer", who);
  MI355_REQUIRE(a->M >= 0 && a->N > 0 && a->K > 0, "%s: bad shape M=%ld N=%ld K=%ld", who,
                (long)a->M, (long)a->N, (long)a->K);
  MI355_REQUIRE(a->M < (1ll << 31) && a->N < (1ll << 31) && a->K < (1ll << 31),
                "%s: dimension exceeds int32", who);
  MI355_REQUIRE((a->gate == nullptr) == (a->residual == nullptr),
                "%s: gate and residual must be given together", who);
  if (a->gate) MI355_REQUIRE(a->gate_rows > 0, "%s: gate_rows must be positive", who);
  MI355_REQUIRE(a->act == MI355_ACT_NONE || a->act == MI355_ACT_GELU_TANH, "%s: unknown act %d",
                who, a->act);
  MI355_REQUIRE(a->f16_begin % 32 == 0 && a->f16_end % 32 == 0 && a->f16_begin >= 0 &&
                    a->f16_end >= a->f16_begin && a->f16_end <= a->N,
                "%s: f16 column range must be 32-aligned and inside [0, N]", who);
  MI355_REQUIRE(a->f16_begin == a->f16_end || (!a->gate && (a->fp8 || a->M > 16)),
                "%s: f16 output columns are not available with gate/residual or on the small-M path",
                who);
  MI355_REQUIRE(!a->out_f16 || (a->fp8 && !a->gate && !a->residual && a->act == MI355_ACT_NONE &&
                                a->f16_begin == a->f16_end),
                "%s: half output is built for the plain fp8 GEMMs (no activation, gate, residual "
                "or half-column range)", who);
  return MI355_OK;
}

// validation of the MFMA path's operands + translation into kernel parameters
int build_params(const mi355_linear_args* a, GemmParams& p) {
  const int esz = a->fp8 ? 1 : 2;
  const int64_t kbytes = a->K * esz;
  MI355_REQUIRE(kbytes % 128 == 0, "linear: K (%ld) must be a multiple of %d", (long)a->K,
                128 / esz);
  MI355_REQUIRE((a->lda * esz) % 16 == 0 && (a->ldw * esz) % 16 == 0,
                "linear: operand row strides must be 16-byte multiples");
  MI355_REQUIRE(((uintptr_t)a->a % 16) == 0 && ((uintptr_t)a->w % 16) == 0,
                "linear: operands must be 16-byte aligned");
  MI355_REQUIRE(a->N % 8 == 0, "linear: N (%ld) must be a multiple of 8", (long)a->N);
  MI355_REQUIRE(a->ldo % 8 == 0 && ((uintptr_t)a->out % 16) == 0 &&
                    (!a->gate || (a->ldg % 8 == 0 && ((uintptr_t)a->gate % 16) == 0 &&
                                  ((uintptr_t)a->residual % 16) == 0)),
                "linear: output/gate/residual rows must be 16-byte aligned");
  MI355_REQUIRE(!a->fp8 || a->block_scale || ((uintptr_t)a->sb % 16) == 0,
                "fp8 linear: scales_b must be 16-byte aligned");
  MI355_REQUIRE(!a->bias || ((uintptr_t)a->bias % 8) == 0, "linear: bias must be 8-byte aligned");
  if (a->fp8) MI355_REQUIRE(a->sa && a->sb, "fp8 linear: scales missing");
  MI355_REQUIRE(!a->block_scale || (a->fp8 && a->K % 128 == 0),
                "linear: block scales need the fp8 path and K %% 128 == 0");
  // the epilogues keep the mapped output row of a store as a 32-bit number
  if (a->o_map.rows_per_batch > 0 && a->M > 0) {
    const int64_t last = ((a->M - 1) / a->o_map.rows_per_batch) * a->o_map.batch_rows +
                         a->o_map.row_offset + a->o_map.rows_per_batch - 1;
    MI355_REQUIRE(a->o_map.batch_rows >= 0 && a->o_map.row_offset >= 0 && last < (1ll << 31),
                  "linear: output row map reaches row %ld, beyond 2^31", (long)last);
  }
  p.a = (const unsigned char*)a->a;
  p.w = (const unsigned char*)a->w;
  p.sa = a->sa;
  p.sb = a->sb;
  p.bias = (const unsigned short*)a->bias;
  p.out = (unsigned short*)a->out;
  p.residual = (const unsigned short*)a->residual;
  p.gate = (const unsigned short*)a->gate;
  p.M = (int)a->M;
  p.N = (int)a->N;
  p.ktiles = (int)(kbytes / 128);
  p.lda_b = a->lda * esz;
  p.ldw_b = a->ldw * esz;
  p.ldo = a->ldo;
  p.ldg = a->ldg;
  p.gate_rows = a->gate_rows > 0 ? a->gate_rows : 1;
  p.a_map = a->a_map;
  p.o_map = a->o_map;
  p.act = a->act;
  p.f16_begin = (int)a->f16_begin;
  p.f16_end = (int)a->f16_end;
  p.f16_counters = nullptr;
  if (p.f16_end > p.f16_begin) {
    p.f16_counters = mi355_f16_range_counters();
    if (!p.f16_counters) return MI355_EHIP;
  }
  p.sa_sm = p.sa_sk = p.sb_sk = p.sb_sn = 0;
  if (a->block_scale) {
    p.sa_sm = a->sa_sm;
    p.sa_sk = a->sa_sk;
    p.sb_sk = a->sb_sk;
    p.sb_sn = a->sb_sn;
  }
  p.group = ((a->N + 255) / 256) <= 16 ? 2 : 8;
  p.out_f16 = a->out_f16 ? 1 : 0;
  return MI355_OK;
}

// the ring kernel addresses its operands with 32-bit byte offsets from their bases
bool ring_offsets_fit(const GemmParams& p) {
  int64_t last_row = p.M - 1;
  if (p.a_map.rows_per_batch > 0)
    last_row = (last_row / p.a_map.rows_per_batch) * p.a_map.batch_rows + p.a_map.row_offset +
               p.a_map.rows_per_batch - 1;
  return (last_row + 1) * p.lda_b < (1ll << 32) && (int64_t)p.N * p.ldw_b < (1ll << 32);
}

int dispatch_gemm(const GemmParams& p, bool fp8, bool block_scale, bool big, hipStream_t st) {
  if (!big && !block_scale && !p.out_f16 && use_narrow_tile(p.M, p.N))
    return fp8 ? launch_gemm<1, 4, 1, 1, 2, false, false, true>(p, st)
               : launch_gemm<0, 4, 1, 1, 2, false, false, true, true>(p, st);
  // half output is an interface-compatibility mode, not a hot path: one tile class serves it
  if (p.out_f16)
    return block_scale ? launch_gemm<1, 2, 2, 2, 2, true, true>(p, st)
                       : launch_gemm<1, 2, 2, 2, 2, false, true>(p, st);
  if (block_scale)
    return big ? launch_gemm<1, 4, 2, 2, 4, true>(p, st) : launch_gemm<1, 2, 2, 2, 2, true>(p, st);
  // The plain forms run on the 16x16 MFMA shapes (v_mfma_f32_16x16x32_bf16,
  // v_mfma_scale_f32_16x16x128_f8f6f4): same bits as the 32x32 shapes (measured over every
  // element of K = 15360 outputs) and 2-6 % (fp8) / 5-8 % (bf16) less time on the Flux shapes --
  // the chip is power-capped under these kernels and holds a higher clock with them.  bf16 adds
  // the software-pipelined K loop (same bits, 3-4 % less time).
  if (fp8) {
    // big tiles through the five-slot ring (same bits as the two-stage kernel, which remains for
    // a single K tile): 3-5 % less time per GEMM, whole step 192.2 -> 188.5 ms in a same-box A/B
    if (big && p.ktiles >= 2 && ring_offsets_fit(p)) return launch_gemm_ring(p, st);
    return big ? launch_gemm<1, 4, 2, 2, 4, false, false, true>(p, st)
               : launch_gemm<1, 2, 2, 2, 2, false, false, true>(p, st);
  }
  return big ? launch_gemm<0, 4, 2, 2, 4, false, false, true, true>(p, st)
             : launch_gemm<0, 2, 2, 2, 2, false, false, true, true>(p, st);
}

int launch_small_m(const mi355_linear_args* a, hipStream_t st) {
  MI355_REQUIRE(a->K % 8 == 0 && a->lda % 8 == 0 && a->ldw % 8 == 0,
                "linear_bf16: K and strides must be multiples of 8");
  MI355_REQUIRE(((uintptr_t)a->a % 16) == 0 && ((uintptr_t)a->w % 16) == 0,
                "linear_bf16: operands must be 16-byte aligned");
  // four channels per wave once there are enough channels to fill the chip that way
  const int ch = a->N >= 16384 ? 4 : 1;
  dim3 grid((unsigned)((a->N + 4 * ch - 1) / (4 * ch))), block(256);
#define SK(MM)                                                                                     \
  do {                                                                                             \
    if (ch == 4)                                                                                   \
      hipLaunchKernelGGL((skinny_linear_kernel<MM, 4>), grid, block, 0, st,                        \
                         (const unsigned short*)a->a, (const unsigned short*)a->w,                 \
                         (const unsigned short*)a->bias, (unsigned short*)a->out, (int)a->M,       \
                         (int)a->N, (int)a->K, a->lda, a->ldw, a->ldo, a->act);                    \
    else                                                                                           \
      hipLaunchKernelGGL((skinny_linear_kernel<MM, 1>), grid, block, 0, st,                        \
                         (const unsigned short*)a->a, (const unsigned short*)a->w,                 \
                         (const unsigned short*)a->bias, (unsigned short*)a->out, (int)a->M,       \
                         (int)a->N, (int)a->K, a->lda, a->ldw, a->ldo, a->act);                    \
  } while (0)
  if (a->M <= 1) SK(1);
  else if (a->M <= 2) SK(2);
  else if (a->M <= 4) SK(4);
  else if (a->M <= 8) SK(8);
  else SK(16);
#undef SK
  MI355_CHECK_LAUNCH("skinny_linear");
  return MI355_OK;
}

}  // namespace

extern "C" int mi355_linear_ex(const mi355_linear_args* a, void* stream) {
  MI355_REQUIRE(a != nullptr, "linear_ex: null args");
  int rc = check_common(a, "linear_ex");
  if (rc) return rc;
  if (a->M == 0) return MI355_OK;
  hipStream_t st = (hipStream_t)stream;
  if (!a->fp8 && a->M <= 16 && a->a_map.rows_per_batch == 0 && a->o_map.rows_per_batch == 0 &&
      !a->gate)
    return launch_small_m(a, st);
  GemmParams p;
  rc = build_params(a, p);
  if (rc) return rc;
  // Tile choice: 256x256 (8 waves) when the grid still covers the chip, else 128x128.
  const bool big = use_big_tile(a->M, a->N, a->fp8 != 0);
  return dispatch_gemm(p, a->fp8 != 0, a->block_scale != 0, big, st);
}

extern "C" int mi355_fp8_scaled_mm(const void* a, const void* b, const float* sa, const float* sb,
                                   const void* bias, void* out, int64_t M, int64_t N, int64_t K,
                                   int64_t lda, int64_t ldb, int64_t ldo, void* stream) {
  mi355_linear_args g = {};
  g.a = a;
  g.w = b;
  g.sa = sa;
  g.sb = sb;
  g.bias = bias;
  g.out = out;
  g.M = M;
  g.N = N;
  g.K = K;
  g.lda = lda;
  g.ldw = ldb;
  g.ldo = ldo;
  g.fp8 = 1;
  return mi355_linear_ex(&g, stream);
}

extern "C" int mi355_linear_bf16(const void* x, const void* w, const void* bias, void* out,
                                 int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldw,
                                 int64_t ldo, int act, void* stream) {
  mi355_linear_args g = {};
  g.a = x;
  g.w = w;
  g.bias = bias;
  g.out = out;
  g.M = M;
  g.N = N;
  g.K = K;
  g.lda = ldx;
  g.ldw = ldw;
  g.ldo = ldo;
  g.act = act;
  return mi355_linear_ex(&g, stream);
}

extern "C" int mi355_fp8_blockwise_scaled_mm(const void* a, const void* b, const float* sa,
                                             const float* sb, void* out, int64_t M, int64_t N,
                                             int64_t K, int64_t lda, int64_t ldb, int64_t ldo,
                                             int64_t sa_sm, int64_t sa_sk, int64_t sb_sk,
                                             int64_t sb_sn, void* stream) {
  MI355_REQUIRE(M == 0 || (a && b && sa && sb && out), "fp8_blockwise_scaled_mm: null pointer");
  MI355_REQUIRE(M >= 0 && N > 0 && K > 0 && M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31),
                "fp8_blockwise_scaled_mm: bad shape");
  MI355_REQUIRE(K % 128 == 0, "fp8_blockwise_scaled_mm: K (%ld) must be a multiple of 128",
                (long)K);
  MI355_REQUIRE(lda % 16 == 0 && ldb % 16 == 0 && ((uintptr_t)a % 16) == 0 &&
                    ((uintptr_t)b % 16) == 0,
                "fp8_blockwise_scaled_mm: operands must be 16-byte aligned");
  MI355_REQUIRE(N % 8 == 0 && ldo % 8 == 0 && ((uintptr_t)out % 16) == 0,
                "fp8_blockwise_scaled_mm: N and output rows must be 16-byte multiples");
  if (M == 0) return MI355_OK;
  mi355_linear_args g = {};
  g.a = a;
  g.w = b;
  g.sa = sa;
  g.sb = sb;
  g.out = out;
  g.M = M;
  g.N = N;
  g.K = K;
  g.lda = lda;
  g.ldw = ldb;
  g.ldo = ldo;
  g.fp8 = 1;
  g.block_scale = 1;
  g.sa_sm = sa_sm;
  g.sa_sk = sa_sk;
  g.sb_sk = sb_sk;
  g.sb_sn = sb_sn;
  return mi355_linear_ex(&g, stream);
}
