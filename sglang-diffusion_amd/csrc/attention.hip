// Flash-attention forward for gfx950: dense softmax(Q K^T * scale) V, bf16
// in/out, fp32 scores/softmax/accumulation, online softmax over 64-key tiles.
//
// Shape of the computation (one workgroup = 4 or 8 wavefronts = 128 or 256 query rows of
// one (batch, head), sharing the K/V tiles in LDS; each wavefront owns 32 query rows):
//   S^T[key][q] = mfma_32x32x16(A = K tile fragment, B = Q fragment)
//       -> the query sits on the LANE, the keys in registers, so the row max /
//          row sum are in-register reductions plus ONE exchange between lane l
//          and lane l+32; no LDS round trip for softmax.
//   O^T[dv][q] += mfma_32x32x16(A = V^T fragment, B = P^T)
//       -> P^T is the S^T accumulator itself (converted to bf16 in place): an
//          accumulator tile is a valid B operand of the next MFMA with no lane
//          movement; m, l and the O rescale all stay lane-local.
// Memory: Q straight to registers (coalesced 16-byte loads, once); K and V
// tiles HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4), double buffered,
// one barrier per tile.  K is read row-wise (ds_read_b128) with the chunk
// index XORed by (key & 15); V is read column-wise with the hardware transpose
// read ds_read_b64_tr_b16 from a row-major image whose chunk index is XORed by
// ((key&3)<<2 | (key>>2)&3): both reads are bank-conflict free.  LDS-DMA
// writes lane-linearly, so the same XOR is applied to each lane's SOURCE
// address.
#include <stdlib.h>

#include <type_traits>

#include <atomic>

#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 hw_f16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

struct AttnParams {
  const unsigned short* q;
  const unsigned short* k;
  const unsigned short* v;
  unsigned short* o;
  float* lse;
  int B, Sq, Sk, H, Hk;
  int64_t q_sb, q_ss, q_sh, k_sb, k_ss, k_sh, v_sb, v_ss, v_sh, o_sb, o_ss, o_sh;
  float scale_log2;  // softmax_scale * log2(e)
  // optional transform of Q while it is loaded: per-head RMSNorm, interleaved RoPE
  const float* qw0;
  const float* qw1;
  int q_split;
  float q_eps;
  const float* qcos;
  const float* qsin;
  int ablate;  // timing experiments only (MI355_ATTN_ABLATE): bit 0 = no K/V refills after the
               // first tiles, bit 1 = per-phase clocks of workgroup 0 into lse (two-phase kernel)
};

__device__ __forceinline__ int k_swz(int key) { return key & 15; }
__device__ __forceinline__ int v_swz(int key) { return ((key & 3) << 2) | ((key >> 2) & 3); }

// Q fragments of one wave: B operand, lane -> query row q0 + r, d = 16s + 8hh + 0..7
template <int D, int KSTEPS>
__device__ __forceinline__ void load_q_fragments(const AttnParams& p, int b, int head, int q0, int r,
                                                 int hh, bf16x8 (&qf)[KSTEPS]) {
  int qrow = q0 + r;
  if (qrow >= p.Sq) qrow = p.Sq - 1;
  const unsigned short* qp = p.q + b * p.q_sb + (int64_t)qrow * p.q_ss + head * p.q_sh;
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s + 8 * hh);
  if (p.qw1 || p.qcos) {
    // per-head RMSNorm and interleaved RoPE of the query while it is loaded.  Same arithmetic and
    // summation order as qk_norm_rope_kernel (elementwise.hip), where 16 (or 8) lanes hold one
    // row in 8-element chunks: here chunk 2s + hh of the row sits in step s of this lane and
    // chunk 2s + (1 - hh) in the lane 32 away, so the butterfly over chunk-index bits 3, 2, 1
    // runs inside the lane and bit 0 is one exchange.
    float f[KSTEPS][8];
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) f[s][j] = (float)qf[s][j];
    const float* wsel = qrow < p.q_split ? p.qw0 : p.qw1;
    if (wsel) {
      float c2[KSTEPS];
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) {
        c2[s] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) c2[s] += f[s][j] * f[s][j];
      }
#pragma unroll
      for (int w = KSTEPS / 2; w >= 1; w >>= 1)
#pragma unroll
        for (int s = 0; s < w; ++s) c2[s] = c2[s] + c2[s + w];
      const float sq = c2[0] + __shfl_xor(c2[0], 32, 64);
      const float rs = rsqrtf(sq / (float)D + p.q_eps);
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) {
        const f32x4 wa = reinterpret_cast<const f32x4*>(wsel)[2 * (2 * s + hh)];
        const f32x4 wb = reinterpret_cast<const f32x4*>(wsel)[2 * (2 * s + hh) + 1];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f[s][j] = rbf((f[s][j] * rs) * wa[j]);
          f[s][4 + j] = rbf((f[s][4 + j] * rs) * wb[j]);
        }
      }
    }
    if (p.qcos) {
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) {
        const f32x4 cs = reinterpret_cast<const f32x4*>(p.qcos + (int64_t)qrow * (D / 2))[2 * s + hh];
        const f32x4 sn = reinterpret_cast<const f32x4*>(p.qsin + (int64_t)qrow * (D / 2))[2 * s + hh];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float x1 = f[s][2 * j], x2 = f[s][2 * j + 1];
          f[s][2 * j] = fmaf(-x2, sn[j], __fmul_rn(x1, cs[j]));
          f[s][2 * j + 1] = fmaf(x1, sn[j], __fmul_rn(x2, cs[j]));
        }
      }
    }
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[s][j] = (__bf16)f[s][j];
  }
}

// D = head size (128 or 64).  KV tile = 64 keys.
// F16: V holds IEEE half values and P is rounded to half (11 significant bits instead of
// bf16's 8) before the PV product; Q, K and the output stay bf16.
// ASMV: the V fragment reads as inline asm (see the P.V phase).
template <int D, bool TRREAD, bool F16, int NW = 4, bool ASMV = false>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void attn_fwd_kernel(AttnParams p) {
  constexpr int KV = 64;
  constexpr int ROWB = D * 2;                // bytes per K/V row
  constexpr int CPR = ROWB / 16;             // 16-B chunks per row (16 or 8)
  constexpr int TILE_BYTES = KV * ROWB;      // one K or V tile
  constexpr int STAGE_BYTES = 2 * TILE_BYTES;
  constexpr int PIECES = 2 * TILE_BYTES / 1024 / NW;  // 1-KiB DMA pieces per wave per stage
  constexpr int KSTEPS = D / 16;             // QK^T k-steps
  constexpr int DVB = D / 32;                // 32-wide dv blocks
  static_assert(D == 128 || D == 64, "head size");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31;
  const int hh = lane >> 5;

  // grid: x = q block (fastest), y = head, z = batch.  Workgroups are dealt to the 8 XCDs
  // round robin by their linear id, and every XCD has its own L2: remap the ids so that each
  // XCD owns a CONTIGUOUS range of (batch, head, q block) work, i.e. its ~64 resident
  // workgroups stream the K/V of two heads instead of fourteen and re-reads hit its L2.
  // Bijective for any grid size.
  int qb, head, b;
  {
    const unsigned gx = gridDim.x, gy = gridDim.y;
    const unsigned total = gx * gy * gridDim.z;
    const unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned per = total >> 3, rem = total & 7, xcd = lin & 7;
    const unsigned wid =
        (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + (lin >> 3);
    qb = (int)(wid % gx);
    const unsigned bh = wid / gx;
    head = (int)(bh % gy);
    b = (int)(bh / gy);
  }
  const int hk = head / (p.H / p.Hk);
  const int q0 = qb * (NW * 32) + wave * 32;

  bf16x8 qf[KSTEPS];
  load_q_fragments<D, KSTEPS>(p, b, head, q0, r, hh, qf);

  // ---- LDS-DMA sources: piece = 1 KiB = 1024/ROWB rows; [K tile | V tile] ----
  // A full tile is fetched as (wave-uniform tile base) + (loop-invariant 32-bit lane offset):
  // no per-lane address arithmetic inside the loop.  Only the last, partial tile clamps rows.
  const unsigned char* kbase =
      reinterpret_cast<const unsigned char*>(p.k + b * p.k_sb + hk * p.k_sh);
  const unsigned char* vbase =
      reinterpret_cast<const unsigned char*>(p.v + b * p.v_sb + hk * p.v_sh);
  static_assert(PIECES % 2 == 0, "first half of a wave's pieces is K, second half V");
  int prow[PIECES];
  int pchunk[PIECES];
  unsigned int voff[PIECES];
#pragma unroll
  for (int i = 0; i < PIECES; ++i) {
    const int piece = wave + NW * i;             // 0 .. 2*TILE_BYTES/1024-1
    const int lin = piece * 64 + lane;          // 16-B slot index in [K|V] stage image
    const int slot = lin % (TILE_BYTES / 16);   // slot within its tile
    const bool isv = i >= PIECES / 2;           // == (lin >= TILE_BYTES / 16)
    const int row = slot / CPR;
    const int pc = slot % CPR;
    prow[i] = row;
    pchunk[i] = isv ? (pc ^ (v_swz(row) & (CPR - 1))) : (pc ^ (k_swz(row) & (CPR - 1)));
    voff[i] = (unsigned int)(row * (isv ? p.v_ss : p.k_ss) * 2 + pchunk[i] * 16);
  }
  const int64_t k_tile_step = (int64_t)KV * p.k_ss * 2, v_tile_step = (int64_t)KV * p.v_ss * 2;

  auto stage = [&](int buf, int k0) {
    if (k0 + KV <= p.Sk) {
      const unsigned char* kt0 = kbase + (int64_t)(k0 / KV) * k_tile_step;  // wave-uniform
      const unsigned char* vt0 = vbase + (int64_t)(k0 / KV) * v_tile_step;
#pragma unroll
      for (int i = 0; i < PIECES; ++i) {
        const int piece = wave + NW * i;
        const unsigned char* src = (i >= PIECES / 2 ? vt0 : kt0) + voff[i];
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src,
                                         (lds_ptr_t)(smem + buf * STAGE_BYTES + piece * 1024), 16,
                                         0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < PIECES; ++i) {
        const int piece = wave + NW * i;
        int key = k0 + prow[i];
        if (key >= p.Sk) key = p.Sk - 1;
        const unsigned char* src = i >= PIECES / 2
                                       ? vbase + ((int64_t)key * p.v_ss * 2 + pchunk[i] * 16)
                                       : kbase + ((int64_t)key * p.k_ss * 2 + pchunk[i] * 16);
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src,
                                         (lds_ptr_t)(smem + buf * STAGE_BYTES + piece * 1024), 16,
                                         0, 0);
      }
    }
  };

  // ---- per-lane LDS read offsets ----------------------------------------------
  // K row read (A operand): row = 32*kb + r, chunk = 2s + hh
  int koff[KSTEPS];
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s)
    koff[s] = r * ROWB + (((2 * s + hh) ^ (k_swz(r) & (CPR - 1))) << 4);
  // V transposed read (A operand of PV): 16-lane group g2 = (lane>>4)&1 covers dv
  // columns 16*g2..16*g2+15 of a 32-wide dv block; lane (4*qq+pp) of the group
  // addresses row (key) r0+qq, columns 4pp..4pp+3.  r0 = 16*st + 4*hh (+8 for the
  // second read) so that element j of this lane is key 16st + 8(j>>2) + 4hh + (j&3),
  // the order in which the S^T accumulator registers hold P.
  const int li = lane & 15, qq = li >> 2, pp = li & 3, g2 = (lane >> 4) & 1;
  // Loop-invariant lane part of the V read address, per dv block and per 8-key half: the
  // swizzle term v_swz(key) only sees key bits 0..3 = (half, hh, qq), so the (kb, st) part of
  // the key is a plain constant that goes into the instruction's immediate offset.
  int vlane[DVB][2];
#pragma unroll
  for (int d = 0; d < DVB; ++d)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int key = half * 8 + 4 * hh + qq;
      const int col = d * 32 + g2 * 16 + 4 * pp;
      vlane[d][half] =
          key * ROWB + ((((col >> 3) ^ (v_swz(key) & (CPR - 1)))) << 4) + ((col & 7) << 1);
    }

  f32x16 oacc[DVB];
#pragma unroll
  for (int d = 0; d < DVB; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
  float m_run = -INFINITY;  // running max of RAW scores (scale applied in the exponent)
  float l_run = 0.f;        // this lane's partial row sum (its 32 of every 64 keys)
  const float c = p.scale_log2;

  const int ntiles = (p.Sk + KV - 1) / KV;
  // 8-wave workgroups: waves w and w + 4 share a SIMD and the later-dispatched one loses every
  // arbitration at equal priority.  Raising waves 4..7 once, with no per-phase flips, measured
  // 0.3-0.5 % less time than flipping the priority around every MFMA phase (same bits).
  if (NW == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);
  stage(0, 0);
  // One KV tile out of LDS buffer BUF (compile time, so every LDS address below is a
  // loop-invariant lane offset plus an immediate).
  auto tile_body = [&](auto bufc, int t) {
    constexpr int cur = decltype(bufc)::value;
    const int k0 = t * KV;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < ntiles && !(p.ablate == 1 && t > 0)) stage(cur ^ 1, k0 + KV);
    const unsigned char* kt = smem + cur * STAGE_BYTES;
    const unsigned char* vt = kt + TILE_BYTES;

    // ---- S^T = K Q^T : two 32-key blocks ------------------------------------
    // The two 32-key chains are interleaved and every K fragment is read one step before the
    // MFMA that consumes it, so neither the LDS latency nor the accumulator dependency of a
    // chain is exposed.
    f32x16 sacc[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc[kb][e] = 0.f;
    // 4-wave workgroups: MFMA phases run at raised wave priority -- when the SIMD's other wave is
    // in its softmax, the matrix unit is fed first (about 1 %).  8-wave workgroups use the static
    // priority set before the loop instead.
    if constexpr (NW != 8) __builtin_amdgcn_s_setprio(2);
    if constexpr (ASMV) {
      // K fragments by inline asm, two steps ahead, waits counted by hand (the compiler's own
      // placement waits with lgkmcnt(0) right behind the reads it has just issued)
      const unsigned lds0k = (unsigned)(uintptr_t)(lds_ptr_t)smem;
      bf16x8 kf[3][2];  // [step % 3][kb]
#define MI355_K_READ(S)                                                                          \
  _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                               \
      asm volatile("ds_read_b128 %0, %1 offset:%2"                                               \
                   : "=v"(kf[(S) % 3][kb])                                                       \
                   : "v"(lds0k + (unsigned)koff[S]), "n"(cur * STAGE_BYTES + kb * 32 * ROWB)     \
                   : "memory")
      MI355_K_READ(0);
      MI355_K_READ(1);
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) {
        if (s + 2 < KSTEPS) MI355_K_READ(s + 2);
        const int later = (s + 2 < KSTEPS ? 2 : KSTEPS - 1 - s) * 2;
        if (later == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        else if (later == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
          sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s % 3][kb], qf[s], sacc[kb], 0, 0, 0);
      }
#undef MI355_K_READ
    } else {
      bf16x8 kf[2][2];  // [parity of step][kb]
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
        kf[0][kb] = *reinterpret_cast<const bf16x8*>(kt + kb * 32 * ROWB + koff[0]);
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) {
        if (s + 1 < KSTEPS) {
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
            kf[(s + 1) & 1][kb] =
                *reinterpret_cast<const bf16x8*>(kt + kb * 32 * ROWB + koff[s + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the reads of step s+1 ahead of step s's MFMAs
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
          sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s & 1][kb], qf[s], sacc[kb], 0, 0, 0);
      }
    }
    if constexpr (NW != 8) __builtin_amdgcn_s_setprio(0);
    // (ASMV) the P.V phase's transposed V reads, see there.  (Requesting the first fragments
    // here, ahead of the softmax, measured slower than requesting them where P.V starts -- and,
    // once the softmax held no LDS exchange any more, the same to 0.2-0.5 %.)
    constexpr int PV_STEPS = 4 * DVB;  // (kb, st, d) steps of the P.V phase, one MFMA each
    constexpr int PV_AHEAD = 3;        // steps whose fragments are in flight ahead of the MFMA
    u32x2 tv[ASMV ? PV_STEPS : 1][2];
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    // (spelled as a macro: inline-asm operands inside a nested generic lambda do not compile)
#define MI355_V_STEP_READ(II)                                                                     \
  do {                                                                                            \
    const int kb_ = (II) / (2 * DVB), st_ = ((II) / DVB) % 2, d_ = (II) % DVB;                    \
    _Pragma("unroll") for (int half = 0; half < 2; ++half)                                        \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2"                                        \
                     : "=v"(tv[II][half])                                                         \
                     : "v"(lds0 + (unsigned)vlane[d_][half]),                                     \
                       "n"(cur * STAGE_BYTES + TILE_BYTES + (kb_ * 32 + st_ * 16) * ROWB)         \
                     : "memory");                                                                 \
  } while (0)
    // mask keys past the end (last tile only; wave-uniform branch)
    if (k0 + KV > p.Sk) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = k0 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
          if (key >= p.Sk) sacc[kb][e] = -INFINITY;
        }
    }
    // ---- online softmax (query on the lane) -----------------------------------
    // Lazy reference point: the exponentials are taken against a reference m_run that only has
    // to be NEAR the row maximum (numerator and denominator share it, the softmax value does not
    // depend on it).  They are formed against the reference as it stands; it moves -- to the true
    // row maximum, with O and l rescaled and the tile's exponentials redone -- only when a lane's
    // sum of its 32 values is not <= LAZY_SUM = 2^13 (so every p that goes on is <= 2^13, which
    // fp32 sums and half / bf16 probabilities hold easily; inf and NaN, i.e. the first tile with
    // m_run = -inf, fail the test as well).  The row maximum itself (17 vector instructions per
    // tile, and an LDS exchange between the two lanes of a row) is only computed when that happens.
    // (Summing the ROUNDED half-precision probabilities two at a time with v_dot2c_f32_f16 against
    // {1, 1} -- 16 instructions for the 32 adds -- measured 2.5 % slower: 906 -> 929 us.)
    constexpr float LAZY_SUM = 8192.0f;
    float pe[2][16];
    float rs;
    auto exponentials = [&](float mc) {
      rs = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          pe[kb][e] = __builtin_amdgcn_exp2f(fmaf(sacc[kb][e], c, -mc));
          rs += pe[kb][e];
        }
    };
    exponentials(m_run * c);
    if (__builtin_amdgcn_ballot_w64(!(rs <= LAZY_SUM)) != 0) {
      float mx = sacc[0][0];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) mx = fmaxf(mx, sacc[kb][e]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < DVB; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[d][e] *= alpha;
      exponentials(m_run * c);
    }
    typedef typename std::conditional<F16, f16x8, bf16x8>::type pfrag_t;
    pfrag_t pf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if constexpr (F16) {
            pf[kb][st][j] = (_Float16)pe[kb][8 * st + j];
          } else {
            pf[kb][st][j] = (__bf16)pe[kb][8 * st + j];
          }
        }
    l_run += rs;
    if constexpr (NW != 8) __builtin_amdgcn_s_setprio(2);

    // ---- O^T += V^T P^T ---------------------------------------------------------
    if constexpr (TRREAD && ASMV) {
      // The transposed reads as inline asm.  Through the builtin, the compiler cannot tell that
      // these reads do not touch the stage the LDS-DMA of the NEXT tile is filling, and puts
      // `s_waitcnt vmcnt(0)` in front of the first of them: the refill issued at the top of the
      // tile then has to land before P.V may start instead of before the next tile does.  In
      // asm the reads are invisible to that analysis; their waits are counted by hand (LDS
      // operations return in order; no scalar loads are issued inside the tile loop).
      constexpr int NIT = PV_STEPS, AHEAD = PV_AHEAD;
#pragma unroll
      for (int ii = 0; ii < AHEAD; ++ii) MI355_V_STEP_READ(ii);
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        if (it + AHEAD < NIT) MI355_V_STEP_READ(it + AHEAD);
        // reads of the steps it + 1 .. min(it + AHEAD, NIT - 1) may stay in flight
        const int later = (it + AHEAD < NIT ? AHEAD : NIT - 1 - it) * 2;
        if (later == 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        else if (later == 6) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        else if (later == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        else if (later == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        static_assert(AHEAD <= 4, "the waits above");
        __builtin_amdgcn_sched_barrier(0);
        const int kb = it / (2 * DVB), st = (it / DVB) % 2, d = it % DVB;
        if constexpr (F16) {
          const f16x8 vf = __builtin_shufflevector(__builtin_bit_cast(f16x4, tv[it][0]),
                                                   __builtin_bit_cast(f16x4, tv[it][1]), 0, 1, 2, 3,
                                                   4, 5, 6, 7);
          oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kb][st], oacc[d], 0, 0, 0);
        } else {
          const bf16x8 vf = __builtin_shufflevector(__builtin_bit_cast(bf16x4, tv[it][0]),
                                                    __builtin_bit_cast(bf16x4, tv[it][1]), 0, 1, 2, 3,
                                                    4, 5, 6, 7);
          oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kb][st], oacc[d], 0, 0, 0);
        }
      }
#undef MI355_V_STEP_READ
    } else
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int d = 0; d < DVB; ++d) {
          if constexpr (TRREAD) {
            const int off[2] = {vlane[d][0] + (kb * 32 + st * 16) * ROWB,
                                vlane[d][1] + (kb * 32 + st * 16) * ROWB};
            if constexpr (F16) {
              const f16x4 t0 = __builtin_bit_cast(
                  f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                             (__attribute__((address_space(3))) hw_f16x4*)(vt + off[0])));
              const f16x4 t1 = __builtin_bit_cast(
                  f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                             (__attribute__((address_space(3))) hw_f16x4*)(vt + off[1])));
              const f16x8 vf = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
              oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kb][st], oacc[d], 0, 0, 0);
            } else {
              const bf16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                  (__attribute__((address_space(3))) bf16x4*)(vt + off[0]));
              const bf16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                  (__attribute__((address_space(3))) bf16x4*)(vt + off[1]));
              const bf16x8 vf = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
              oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kb][st], oacc[d], 0, 0, 0);
            }
          } else if constexpr (!F16) {
            // debug path: element gathers, same image
            bf16x8 vf;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
              const int key = kb * 32 + st * 16 + (jj >> 2) * 8 + 4 * hh + (jj & 3);
              const int col = d * 32 + r;
              const int off = key * ROWB + ((((col >> 3) ^ (v_swz(key) & (CPR - 1)))) << 4) +
                              ((col & 7) << 1);
              vf[jj] = __builtin_bit_cast(__bf16, *reinterpret_cast<const unsigned short*>(vt + off));
            }
            oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kb][st], oacc[d], 0, 0, 0);
          }
        }
    if constexpr (NW != 8) __builtin_amdgcn_s_setprio(0);
  };
  for (int t = 0; t < ntiles; t += 2) {
    tile_body(std::integral_constant<int, 0>{}, t);
    if (t + 1 < ntiles) tile_body(std::integral_constant<int, 1>{}, t + 1);
  }

  // ---- finish: combine the two half-row sums, normalise, store -------------------
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv_l = 1.0f / l_tot;
  const int qrow = q0 + r;
  const bool q_ok = qrow < p.Sq;
  if (p.lse && q_ok && hh == 0) {
    // natural-log LSE of the scaled scores
    p.lse[((int64_t)b * p.H + head) * p.Sq + qrow] =
        (m_run * c + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f;
  }
  unsigned short* op = p.o + b * p.o_sb + (int64_t)(q_ok ? qrow : 0) * p.o_ss + head * p.o_sh;
#pragma unroll
  for (int d = 0; d < DVB; ++d) {
    unsigned int pk[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      pk[g][0] = pack_bf2(oacc[d][4 * g] * inv_l, oacc[d][4 * g + 1] * inv_l);
      pk[g][1] = pack_bf2(oacc[d][4 * g + 2] * inv_l, oacc[d][4 * g + 3] * inv_l);
    }
#pragma unroll
    for (int g = 0; g < 4; g += 2) {
#pragma unroll
      for (int w2 = 0; w2 < 2; ++w2) {
        auto sw = __builtin_amdgcn_permlane32_swap(pk[g][w2], pk[g + 1][w2], false, false);
        pk[g][w2] = sw[0];
        pk[g + 1][w2] = sw[1];
      }
      if (q_ok) {
        const int dv8 = d * 32 + 8 * (g + hh);
        *reinterpret_cast<u32x4*>(op + dv8) = u32x4{pk[g][0], pk[g][1], pk[g + 1][0], pk[g + 1][1]};
      }
    }
  }
}

// ---- the same computation on the 16x16x32 MFMA shapes ------------------------------------------
// The chip is power-capped under this kernel and holds a higher clock on the 16x16 shapes
// (tools/probes/mfma_shape.hip).  Per wave still 32 query rows and 64-key tiles; what changes is
// who holds what.  With li = lane & 15 and g = lane >> 4:
//   S^T block (kb, jb) = K rows 16 kb .. x Q rows 16 jb ..: lane holds query 16 jb + li and
//       keys 16 kb + 4 g + e (e = 0..3).  A row's 64 scores sit in FOUR lanes, so a lane
//       compares only its own 16 scores with the running reference point; the cross-lane row
//       maximum (two exchanges) is taken only when some lane of the wave has outgrown it.
//   P^T operand of the P.V MFMA over keys 32 kp ..: k index 8 g + j  <->  key
//       32 kp + 16 (j >> 2) + 4 g + (j & 3), i.e. the accumulators of blocks 2 kp and 2 kp + 1 as
//       they stand; the V^T operand is read with the same assignment (two transposed reads of
//       4 keys x 16 columns per 16-lane group).
//   O^T block (d, jb): lane holds query 16 jb + li, columns 16 d + 4 g + e.
// The V image uses its own chunk swizzle, (key & 7) << 1: a transposed read touches two chunks
// in each of eight rows per half-wave, and this spreads them over all sixteen 16-byte slots.
__device__ __forceinline__ int v_swz16(int key) { return (key & 7) << 1; }

// Q fragments (B operand) of one 16-row query block: lane -> row qrow, d = 32 s + 8 g + 0..7.
// Norm / rotation as load_q_fragments: chunk 4 s + g of the row sits in step s of this lane, so
// the butterfly over chunk-index bits 3, 2 runs inside the lane and bits 1, 0 are exchanges
// with the lanes 32 and 16 away.
template <int D, int KS>
__device__ __forceinline__ void load_q_fragments16(const AttnParams& p, int b, int head, int qrow,
                                                   int g, bf16x8 (&qf)[KS]) {
  if (qrow >= p.Sq) qrow = p.Sq - 1;
  const unsigned short* qp = p.q + b * p.q_sb + (int64_t)qrow * p.q_ss + head * p.q_sh;
#pragma unroll
  for (int s = 0; s < KS; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 32 * s + 8 * g);
  if (p.qw1 || p.qcos) {
    float f[KS][8];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) f[s][j] = (float)qf[s][j];
    const float* wsel = qrow < p.q_split ? p.qw0 : p.qw1;
    if (wsel) {
      float c2[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        c2[s] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) c2[s] += f[s][j] * f[s][j];
      }
#pragma unroll
      for (int w = KS / 2; w >= 1; w >>= 1)
#pragma unroll
        for (int s = 0; s < w; ++s) c2[s] = c2[s] + c2[s + w];
      float sq = c2[0] + __shfl_xor(c2[0], 32, 64);
      sq = sq + __shfl_xor(sq, 16, 64);
      const float rs = rsqrtf(sq / (float)D + p.q_eps);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const f32x4 wa = reinterpret_cast<const f32x4*>(wsel)[2 * (4 * s + g)];
        const f32x4 wb = reinterpret_cast<const f32x4*>(wsel)[2 * (4 * s + g) + 1];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f[s][j] = rbf((f[s][j] * rs) * wa[j]);
          f[s][4 + j] = rbf((f[s][4 + j] * rs) * wb[j]);
        }
      }
    }
    if (p.qcos) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const f32x4 cs = reinterpret_cast<const f32x4*>(p.qcos + (int64_t)qrow * (D / 2))[4 * s + g];
        const f32x4 sn = reinterpret_cast<const f32x4*>(p.qsin + (int64_t)qrow * (D / 2))[4 * s + g];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float x1 = f[s][2 * j], x2 = f[s][2 * j + 1];
          f[s][2 * j] = fmaf(-x2, sn[j], __fmul_rn(x1, cs[j]));
          f[s][2 * j + 1] = fmaf(x1, sn[j], __fmul_rn(x2, cs[j]));
        }
      }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[s][j] = (__bf16)f[s][j];
  }
}

template <int D, bool F16, int NW = 4>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void attn_fwd16_kernel(AttnParams p) {
  constexpr int KV = 64;
  constexpr int ROWB = D * 2;
  constexpr int CPR = ROWB / 16;
  constexpr int TILE_BYTES = KV * ROWB;
  constexpr int STAGE_BYTES = 2 * TILE_BYTES;
  constexpr int PIECES = 2 * TILE_BYTES / 1024 / NW;
  constexpr int KS = D / 32;    // QK^T k-steps
  constexpr int DV16 = D / 16;  // 16-wide dv blocks
  static_assert(D == 128, "the V swizzle is laid out for 256-byte rows");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15;
  const int g = lane >> 4;

  // XCD-contiguous work ranges, as attn_fwd_kernel
  int qb, head, b;
  {
    const unsigned gx = gridDim.x, gy = gridDim.y;
    const unsigned total = gx * gy * gridDim.z;
    const unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned per = total >> 3, rem = total & 7, xcd = lin & 7;
    const unsigned wid =
        (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + (lin >> 3);
    qb = (int)(wid % gx);
    const unsigned bh = wid / gx;
    head = (int)(bh % gy);
    b = (int)(bh / gy);
  }
  const int hk = head / (p.H / p.Hk);
  const int q0 = qb * (NW * 32) + wave * 32;

  bf16x8 qf[2][KS];
#pragma unroll
  for (int jb = 0; jb < 2; ++jb) load_q_fragments16<D, KS>(p, b, head, q0 + jb * 16 + li, g, qf[jb]);

  // ---- LDS-DMA sources: [K tile | V tile], 1-KiB pieces --------------------------------------
  const unsigned char* kbase =
      reinterpret_cast<const unsigned char*>(p.k + b * p.k_sb + hk * p.k_sh);
  const unsigned char* vbase =
      reinterpret_cast<const unsigned char*>(p.v + b * p.v_sb + hk * p.v_sh);
  static_assert(PIECES % 2 == 0, "first half of a wave's pieces is K, second half V");
  int prow[PIECES];
  int pchunk[PIECES];
  unsigned int voff[PIECES];
#pragma unroll
  for (int i = 0; i < PIECES; ++i) {
    const int piece = wave + NW * i;
    const int lin = piece * 64 + lane;
    const int slot = lin % (TILE_BYTES / 16);
    const bool isv = i >= PIECES / 2;
    const int row = slot / CPR;
    const int pc = slot % CPR;
    prow[i] = row;
    pchunk[i] = isv ? (pc ^ (v_swz16(row) & (CPR - 1))) : (pc ^ (k_swz(row) & (CPR - 1)));
    voff[i] = (unsigned int)(row * (isv ? p.v_ss : p.k_ss) * 2 + pchunk[i] * 16);
  }
  const int64_t k_tile_step = (int64_t)KV * p.k_ss * 2, v_tile_step = (int64_t)KV * p.v_ss * 2;

  auto stage = [&](int buf, int k0) {
    if (k0 + KV <= p.Sk) {
      const unsigned char* kt0 = kbase + (int64_t)(k0 / KV) * k_tile_step;  // wave-uniform
      const unsigned char* vt0 = vbase + (int64_t)(k0 / KV) * v_tile_step;
#pragma unroll
      for (int i = 0; i < PIECES; ++i) {
        const int piece = wave + NW * i;
        const unsigned char* src = (i >= PIECES / 2 ? vt0 : kt0) + voff[i];
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src,
                                         (lds_ptr_t)(smem + buf * STAGE_BYTES + piece * 1024), 16,
                                         0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < PIECES; ++i) {
        const int piece = wave + NW * i;
        int key = k0 + prow[i];
        if (key >= p.Sk) key = p.Sk - 1;
        const unsigned char* src = i >= PIECES / 2
                                       ? vbase + ((int64_t)key * p.v_ss * 2 + pchunk[i] * 16)
                                       : kbase + ((int64_t)key * p.k_ss * 2 + pchunk[i] * 16);
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src,
                                         (lds_ptr_t)(smem + buf * STAGE_BYTES + piece * 1024), 16,
                                         0, 0);
      }
    }
  };

  // ---- per-lane LDS read offsets ----------------------------------------------------------
  // K row read (A operand): row 16 kb + li, chunk 4 s + g
  int koff[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) koff[s] = li * ROWB + (((4 * s + g) ^ (k_swz(li) & (CPR - 1))) << 4);
  // V transposed read (A operand of P.V): the 16-lane group g reads keys .. + 4 g + qq
  // (qq = li >> 2), columns 16 d + 4 pp .. + 3 (pp = li & 3), and gets back column 16 d + li
  // of those four keys.  The swizzle only sees key bits 0..2, so the (kp, half) part of the key
  // is an immediate offset.
  const int qq = li >> 2, pp = li & 3;
  int vlane[DV16];
#pragma unroll
  for (int d = 0; d < DV16; ++d) {
    const int key = 4 * g + qq;
    vlane[d] = key * ROWB + ((((2 * d + (pp >> 1)) ^ (v_swz16(key) & (CPR - 1)))) << 4) + ((pp & 1) << 3);
  }

  f32x4 oacc[DV16][2];
#pragma unroll
  for (int d = 0; d < DV16; ++d)
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) oacc[d][jb] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run[2] = {-INFINITY, -INFINITY};  // reference point of the exponentials (raw scores)
  float l_run[2] = {0.f, 0.f};              // this lane's partial row sums (its 16 of every 64 keys)
  const float c = p.scale_log2;

  const int ntiles = (p.Sk + KV - 1) / KV;
  stage(0, 0);
  auto tile_body = [&](auto bufc, int t) {
    constexpr int cur = decltype(bufc)::value;
    const int k0 = t * KV;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < ntiles && !(p.ablate == 1 && t > 0)) stage(cur ^ 1, k0 + KV);
    const unsigned char* kt = smem + cur * STAGE_BYTES;
    const unsigned char* vt = kt + TILE_BYTES;

    // ---- S^T = K Q^T : four 16-key blocks x two 16-query blocks -------------------------
    f32x4 sacc[4][2];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) sacc[kb][jb] = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_s_setprio(2);
    {
      bf16x8 kf[2][4];  // [parity of step][kb]
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
        kf[0][kb] = *reinterpret_cast<const bf16x8*>(kt + kb * 16 * ROWB + koff[0]);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if (s + 1 < KS) {
#pragma unroll
          for (int kb = 0; kb < 4; ++kb)
            kf[(s + 1) & 1][kb] =
                *reinterpret_cast<const bf16x8*>(kt + kb * 16 * ROWB + koff[s + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the reads of step s+1 ahead of step s's MFMAs
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int jb = 0; jb < 2; ++jb)
            sacc[kb][jb] =
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[s & 1][kb], qf[jb][s], sacc[kb][jb], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    // mask keys past the end (last tile only; wave-uniform branch)
    if (k0 + KV > p.Sk) {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int key = k0 + kb * 16 + 4 * g + e;
          if (key >= p.Sk) {
            sacc[kb][0][e] = -INFINITY;
            sacc[kb][1][e] = -INFINITY;
          }
        }
    }
    // ---- online softmax ---------------------------------------------------------------
    // Lazy reference point with the sum test (see attn_fwd_kernel): the exponentials are formed
    // against the reference as it stands; only when a lane's sum is not <= LAZY_SUM is the row
    // maximum taken (across the four lanes of a row, so that they move to the same point) and
    // the row's exponentials redone.
    constexpr float LAZY_SUM = 8192.0f;
    typedef typename std::conditional<F16, f16x8, bf16x8>::type pfrag_t;
    pfrag_t pf[2][2];  // [kp][jb]
    float rs[2];
    auto exponentials = [&](auto jbc) {
      constexpr int jb = decltype(jbc)::value;
      const float mc = m_run[jb] * c;
      rs[jb] = 0.f;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float pe = __builtin_amdgcn_exp2f(fmaf(sacc[kb][jb][e], c, -mc));
          rs[jb] += pe;
          if constexpr (F16) pf[kb >> 1][jb][(kb & 1) * 4 + e] = (_Float16)pe;
          else pf[kb >> 1][jb][(kb & 1) * 4 + e] = (__bf16)pe;
        }
    };
    auto move_reference = [&](auto jbc) {
      constexpr int jb = decltype(jbc)::value;
      float mx = sacc[0][jb][0];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int e = 0; e < 4; ++e) mx = fmaxf(mx, sacc[kb][jb][e]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run[jb], mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run[jb] - m_new) * c);
      m_run[jb] = m_new;
      l_run[jb] *= alpha;
#pragma unroll
      for (int d = 0; d < DV16; ++d) oacc[d][jb] *= alpha;
    };
    constexpr std::integral_constant<int, 0> J0{};
    constexpr std::integral_constant<int, 1> J1{};
    exponentials(J0);
    exponentials(J1);
    if (__builtin_amdgcn_ballot_w64(!(rs[0] <= LAZY_SUM) || !(rs[1] <= LAZY_SUM)) != 0) {
      move_reference(J0);
      move_reference(J1);
      exponentials(J0);
      exponentials(J1);
    }
    l_run[0] += rs[0];
    l_run[1] += rs[1];
    __builtin_amdgcn_s_setprio(2);

    // ---- O^T += V^T P^T ---------------------------------------------------------------
#pragma unroll
    for (int kp = 0; kp < 2; ++kp)
#pragma unroll
      for (int d = 0; d < DV16; ++d) {
        const int off0 = vlane[d] + (kp * 32) * ROWB, off1 = vlane[d] + (kp * 32 + 16) * ROWB;
        if constexpr (F16) {
          const f16x4 t0 = __builtin_bit_cast(
              f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                         (__attribute__((address_space(3))) hw_f16x4*)(vt + off0)));
          const f16x4 t1 = __builtin_bit_cast(
              f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                         (__attribute__((address_space(3))) hw_f16x4*)(vt + off1)));
          const f16x8 vf = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
          for (int jb = 0; jb < 2; ++jb)
            oacc[d][jb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[kp][jb], oacc[d][jb], 0, 0, 0);
        } else {
          const bf16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
              (__attribute__((address_space(3))) bf16x4*)(vt + off0));
          const bf16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
              (__attribute__((address_space(3))) bf16x4*)(vt + off1));
          const bf16x8 vf = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
          for (int jb = 0; jb < 2; ++jb)
            oacc[d][jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[kp][jb], oacc[d][jb], 0, 0, 0);
        }
      }
    __builtin_amdgcn_s_setprio(0);
  };
  for (int t = 0; t < ntiles; t += 2) {
    tile_body(std::integral_constant<int, 0>{}, t);
    if (t + 1 < ntiles) tile_body(std::integral_constant<int, 1>{}, t + 1);
  }

  // ---- finish: combine the four partial row sums, normalise, store --------------------------
#pragma unroll
  for (int jb = 0; jb < 2; ++jb) {
    float l_tot = l_run[jb] + __shfl_xor(l_run[jb], 16, 64);
    l_tot = l_tot + __shfl_xor(l_tot, 32, 64);
    const float inv_l = 1.0f / l_tot;
    const int qrow = q0 + jb * 16 + li;
    const bool q_ok = qrow < p.Sq;
    if (p.lse && q_ok && g == 0) {
      p.lse[((int64_t)b * p.H + head) * p.Sq + qrow] =
          (m_run[jb] * c + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f;
    }
    unsigned short* op = p.o + b * p.o_sb + (int64_t)(q_ok ? qrow : 0) * p.o_ss + head * p.o_sh;
    // lanes g and g ^ 1 exchange so that each stores 8 consecutive columns: the even one those
    // of dv block 2 m, the odd one those of block 2 m + 1
#pragma unroll
    for (int m = 0; m < DV16 / 2; ++m) {
      unsigned int a0 = pack_bf2(oacc[2 * m][jb][0] * inv_l, oacc[2 * m][jb][1] * inv_l);
      unsigned int a1 = pack_bf2(oacc[2 * m][jb][2] * inv_l, oacc[2 * m][jb][3] * inv_l);
      unsigned int b0 = pack_bf2(oacc[2 * m + 1][jb][0] * inv_l, oacc[2 * m + 1][jb][1] * inv_l);
      unsigned int b1 = pack_bf2(oacc[2 * m + 1][jb][2] * inv_l, oacc[2 * m + 1][jb][3] * inv_l);
      auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
      auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
      if (q_ok) {
        const int dv8 = (2 * m + (g & 1)) * 16 + 8 * (g >> 1);
        *reinterpret_cast<u32x4*>(op + dv8) = u32x4{s0[0], s1[0], s0[1], s1[1]};
      }
    }
  }
}

// ---- two-phase ("ping-pong") variant: head size 128, 8 waves ----------------------------------
// attn_fwd_kernel runs every wave through  S = K Q^T (MFMA) -> softmax (vector) -> O += V^T P
// (MFMA)  behind one barrier per tile, so the two waves that share a SIMD are always in the SAME
// phase: they queue for the matrix unit together, then for the vector unit together, and each
// unit idles while the other works (measured: a tile costs about the SUM of its MFMA and vector
// time).  Here a wave's iteration is regrouped into an MFMA block [O += V_{j-1}^T P_{j-1}, then
// S_j = K_j Q^T, interleaved] and a vector block [softmax of S_j], two barriers per tile split
// the time into half steps, and waves 4..7 run one half step behind waves 0..3: while one wave
// of a SIMD is in its MFMA block the other is in its vector block.  Same arithmetic in the same
// order as attn_fwd_kernel, so results are identical to the bit.
//   half step 2j   : waves 0-3  MFMA block j          waves 4-7  softmax of tile j-1
//   half step 2j+1 : waves 0-3  softmax of tile j     waves 4-7  MFMA block j
// K_j is read in half steps 2j, 2j+1 and V_j in 2j+2, 2j+3, so K and V each have a two-slot ring:
// at half step 2j the refills for K_{j+1} and V_j are issued (64 KiB of LDS, every offset an
// immediate).
// Measured at B=4, S=4608, H=24 (same box): 1120-1330 us against 930-980 us for attn_fwd_kernel,
// whose waves drift apart inside a tile and overlap on their own; the two workgroup-wide barriers
// per tile cost more than the forced alternation gains.  Kept behind MI355_ATTN_PINGPONG=1 as a
// measured negative result.
template <bool F16>
__global__ __launch_bounds__(512, 2) void attn_fwd_pingpong_kernel(AttnParams p) {
  constexpr int D = 128, NW = 8, KV = 64;
  constexpr int ROWB = D * 2;
  constexpr int CPR = ROWB / 16;
  constexpr int TILE_BYTES = KV * ROWB;
  constexpr int V_RING = 2 * TILE_BYTES;         // LDS: K slot 0 | K slot 1 | V slot 0 | V slot 1
  constexpr int PIECES = TILE_BYTES / 1024 / NW;  // 1-KiB DMA pieces per wave per K (or V) tile
  constexpr int KSTEPS = D / 16;
  constexpr int DVB = D / 32;
  typedef typename std::conditional<F16, f16x8, bf16x8>::type pfrag_t;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31;
  const int hh = lane >> 5;
  // which of a SIMD's two waves this is: a workgroup's waves are dealt to the four SIMDs
  // cyclically, so waves w and w + 4 share one (pairing w with w ^ 1 instead: 1430 us vs 1120)
  const int late = wave >> 2;

  int qb, head, b;
  {
    const unsigned gx = gridDim.x, gy = gridDim.y;
    const unsigned total = gx * gy * gridDim.z;
    const unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned per = total >> 3, rem = total & 7, xcd = lin & 7;
    const unsigned wid =
        (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + (lin >> 3);
    qb = (int)(wid % gx);
    const unsigned bh = wid / gx;
    head = (int)(bh % gy);
    b = (int)(bh / gy);
  }
  const int hk = head / (p.H / p.Hk);
  const int q0 = qb * (NW * 32) + wave * 32;

  bf16x8 qf[KSTEPS];
  load_q_fragments<D, KSTEPS>(p, b, head, q0, r, hh, qf);

  // ---- LDS-DMA sources (as attn_fwd_kernel) -------------------------------------------------
  const unsigned char* kbase =
      reinterpret_cast<const unsigned char*>(p.k + b * p.k_sb + hk * p.k_sh);
  const unsigned char* vbase =
      reinterpret_cast<const unsigned char*>(p.v + b * p.v_sb + hk * p.v_sh);
  // (row, swizzled chunk) of the 16-byte slot this lane fills in piece i of a K or V tile
  auto piece_slot = [&](int i, bool isv, int& row, int& chunk) __attribute__((always_inline)) {
    const int slot = (wave + NW * i) * 64 + lane;
    row = slot / CPR;
    const int pc = slot % CPR;
    chunk = isv ? (pc ^ (v_swz(row) & (CPR - 1))) : (pc ^ (k_swz(row) & (CPR - 1)));
  };
  unsigned int koffg[PIECES], voffg[PIECES];
#pragma unroll
  for (int i = 0; i < PIECES; ++i) {
    int row, chunk;
    piece_slot(i, false, row, chunk);
    koffg[i] = (unsigned int)(row * p.k_ss * 2 + chunk * 16);
    piece_slot(i, true, row, chunk);
    voffg[i] = (unsigned int)(row * p.v_ss * 2 + chunk * 16);
  }
  const int64_t k_tile_step = (int64_t)KV * p.k_ss * 2, v_tile_step = (int64_t)KV * p.v_ss * 2;
  // K (ISV = false) or V tile `t` -> its ring slot t & 1
  auto stage = [&](auto slotc, auto isvc, int t) __attribute__((always_inline)) {
    constexpr int SLOT = decltype(slotc)::value;
    constexpr bool ISV = decltype(isvc)::value;
    unsigned char* dst = smem + (ISV ? V_RING : 0) + SLOT * TILE_BYTES;
    const unsigned char* base = ISV ? vbase : kbase;
    const int k0 = t * KV;
    if (k0 + KV <= p.Sk) {
      const unsigned char* t0 = base + (int64_t)t * (ISV ? v_tile_step : k_tile_step);
#pragma unroll
      for (int i = 0; i < PIECES; ++i)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(t0 + (ISV ? voffg[i] : koffg[i])),
                                         (lds_ptr_t)(dst + (wave + NW * i) * 1024), 16, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < PIECES; ++i) {
        int row, chunk;
        piece_slot(i, ISV, row, chunk);  // the partial last tile only: recomputed, not kept live
        int key = k0 + row;
        if (key >= p.Sk) key = p.Sk - 1;
        const unsigned char* src = base + ((int64_t)key * (ISV ? p.v_ss : p.k_ss) * 2 + chunk * 16);
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(dst + (wave + NW * i) * 1024),
                                         16, 0, 0);
      }
    }
  };

  // ---- per-lane LDS read offsets (as attn_fwd_kernel) ---------------------------------------
  // K row read: row r, chunk (2s + hh) ^ (r & 15).  The k-step only enters through bits 5..7,
  // which the rest of the offset leaves clear: offset(s) = offset(0) ^ (s << 5), one register
  // instead of eight.
  const int koff0 = r * ROWB + ((hh ^ (k_swz(r) & (CPR - 1))) << 4);
  const int li = lane & 15, qq = li >> 2, pp = li & 3, g2 = (lane >> 4) & 1;
  int vlane[DVB][2];
#pragma unroll
  for (int d = 0; d < DVB; ++d)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int key = half * 8 + 4 * hh + qq;
      const int col = d * 32 + g2 * 16 + 4 * pp;
      vlane[d][half] =
          key * ROWB + ((((col >> 3) ^ (v_swz(key) & (CPR - 1)))) << 4) + ((col & 7) << 1);
    }

  f32x16 oacc[DVB];
#pragma unroll
  for (int d = 0; d < DVB; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
  f32x16 sacc[2];    // raw scores of the tile whose MFMA block ran last
  pfrag_t pf[2][2];  // its probabilities after the vector block: [32-key block][16-key step]
  float m_run = -INFINITY;
  float l_run = 0.f;
  const float c = p.scale_log2;
  constexpr float LAZY_SUM = 8192.0f;
  const int ntiles = (p.Sk + KV - 1) / KV;

  auto v_frag = [&](const unsigned char* vt, int key0, int d) __attribute__((always_inline)) {
    const int o0 = vlane[d][0] + key0 * ROWB, o1 = vlane[d][1] + key0 * ROWB;
    if constexpr (F16) {
      const f16x4 t0 = __builtin_bit_cast(
          f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                     (__attribute__((address_space(3))) hw_f16x4*)(vt + o0)));
      const f16x4 t1 = __builtin_bit_cast(
          f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                     (__attribute__((address_space(3))) hw_f16x4*)(vt + o1)));
      return (pfrag_t)__builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
    } else {
      const bf16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
          (__attribute__((address_space(3))) bf16x4*)(vt + o0));
      const bf16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
          (__attribute__((address_space(3))) bf16x4*)(vt + o1));
      return (pfrag_t)__builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };
  auto pv_mfma = [&](const pfrag_t& vf, const pfrag_t& pfr, f32x16& acc) __attribute__((always_inline)) {
    if constexpr (F16) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pfr, acc, 0, 0, 0);
    else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pfr, acc, 0, 0, 0);
  };

  // MFMA block of tile j (K in slot BUF = j & 1): O += V_{j-1}^T P_{j-1} (V in the other slot) and
  // S_j = K_j Q^T, one MFMA of each in turn; fragments are read AHEAD steps before their MFMAs.
  // DO_PV / DO_QK drop one of the two for the first and the last block.
  auto mfma_block = [&](auto bufc, auto do_pv, auto do_qk) __attribute__((always_inline)) {
    constexpr int BUF = decltype(bufc)::value;
    constexpr bool DO_PV = decltype(do_pv)::value, DO_QK = decltype(do_qk)::value;
    const unsigned char* kt = smem + BUF * TILE_BYTES;
    const unsigned char* vt = smem + V_RING + (BUF ^ 1) * TILE_BYTES;
    constexpr int AHEAD = 2, NSTEP = 16;  // 3, 4, 6 steps ahead: spills, no faster
    __builtin_amdgcn_s_setprio(3);
    bf16x8 kf[AHEAD + 1];
    pfrag_t vf[AHEAD + 1];
    // step g: S chain kb = g & 1, k-step g >> 1;  O accumulator d = g & 3 of 16-key step g >> 2
    auto fetch = [&](int g) __attribute__((always_inline)) {
      if constexpr (DO_QK)
        kf[g % (AHEAD + 1)] =
            *reinterpret_cast<const bf16x8*>(kt + (g & 1) * 32 * ROWB + (koff0 ^ ((g >> 1) << 5)));
      if constexpr (DO_PV) vf[g % (AHEAD + 1)] = v_frag(vt, (g >> 2) * 16, g & 3);
    };
    if constexpr (DO_QK) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[kb][e] = 0.f;
    }
#pragma unroll
    for (int g = 0; g < AHEAD; ++g) fetch(g);
#pragma unroll
    for (int g = 0; g < NSTEP; ++g) {
      if (g + AHEAD < NSTEP) fetch(g + AHEAD);
      if constexpr (DO_QK)
        sacc[g & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[g % (AHEAD + 1)], qf[g >> 1],
                                                              sacc[g & 1], 0, 0, 0);
      if constexpr (DO_PV) pv_mfma(vf[g % (AHEAD + 1)], pf[g >> 3][(g >> 2) & 1], oacc[g & 3]);
    }
    // the order the machine scheduler must produce: reads stay AHEAD steps in front
    constexpr int RD = (DO_QK ? 1 : 0) + (DO_PV ? 2 : 0), MF = (DO_QK ? 1 : 0) + (DO_PV ? 1 : 0);
    __builtin_amdgcn_sched_group_barrier(0x100, RD * AHEAD, 0);
#pragma unroll
    for (int g = 0; g < NSTEP - AHEAD; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x100, RD, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, MF, 0);
    }
#pragma unroll
    for (int g = NSTEP - AHEAD; g < NSTEP; ++g) __builtin_amdgcn_sched_group_barrier(0x008, MF, 0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(0);
  };

  // vector block of tile t: online softmax of sacc -> pf, exactly as in attn_fwd_kernel
  auto softmax_tile = [&](int t) __attribute__((always_inline)) {
    const int k0 = t * KV;
    if (k0 + KV > p.Sk) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = k0 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
          if (key >= p.Sk) sacc[kb][e] = -INFINITY;
        }
    }
    // (the sum test of attn_fwd_kernel's lazy reference point, same arithmetic in the same order)
    float pe[2][16];
    float rs;
    auto exponentials = [&](float mc) {
      rs = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          pe[kb][e] = __builtin_amdgcn_exp2f(fmaf(sacc[kb][e], c, -mc));
          rs += pe[kb][e];
        }
    };
    exponentials(m_run * c);
    if (__builtin_amdgcn_ballot_w64(!(rs <= LAZY_SUM)) != 0) {
      float mx = sacc[0][0];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) mx = fmaxf(mx, sacc[kb][e]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < DVB; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[d][e] *= alpha;
      exponentials(m_run * c);
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if constexpr (F16) pf[kb][st][j] = (_Float16)pe[kb][8 * st + j];
          else pf[kb][st][j] = (__bf16)pe[kb][8 * st + j];
        }
    l_run += rs;
  };

  constexpr std::true_type YES{};
  constexpr std::false_type NO{};
  constexpr std::integral_constant<int, 0> S0{};
  constexpr std::integral_constant<int, 1> S1{};
  // timing experiment (MI355_ATTN_ABLATE bit 1): shader clocks spent in MFMA blocks, vector
  // blocks and waiting at barriers, per wave
  unsigned long long tk = 0, t_mfma = 0, t_vec = 0, t_bar = 0;
  auto lap = [&](unsigned long long& acc) __attribute__((always_inline)) {
    if (p.ablate & 2) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      acc += now - tk;
      tk = now;
    }
  };
  auto wait_and_sync = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    lap(t_bar);
  };
  auto sync = [&]() __attribute__((always_inline)) {
    __syncthreads();
    lap(t_bar);
  };
  // the two half steps of tile j (1 <= j < ntiles, J2 = j & 1) for the early / the late waves
  auto tile_steps = [&](auto latec, auto j2c, int j) __attribute__((always_inline)) {
    constexpr bool LATE = decltype(latec)::value;
    constexpr int J2 = decltype(j2c)::value;
    wait_and_sync();  // half step 2j: K_j and V_{j-1} have landed
    if (!(p.ablate == 1 && j > 1)) {
      if (j + 1 < ntiles) stage(std::integral_constant<int, J2 ^ 1>{}, NO, j + 1);
      stage(j2c, YES, j);
    }
    if constexpr (LATE) {
      softmax_tile(j - 1);
      lap(t_vec);
    } else {
      mfma_block(j2c, YES, YES);
      lap(t_mfma);
    }
    sync();  // half step 2j + 1
    if constexpr (LATE) {
      mfma_block(j2c, YES, YES);
      lap(t_mfma);
    } else {
      softmax_tile(j);
      lap(t_vec);
    }
  };
  auto run = [&](auto latec) __attribute__((always_inline)) {
    constexpr bool LATE = decltype(latec)::value;
    // tile 0: nothing to multiply into O yet
    wait_and_sync();
    if (ntiles > 1) stage(S1, NO, 1);
    stage(S0, YES, 0);
    if constexpr (!LATE) mfma_block(S0, NO, YES);
    __syncthreads();
    if (p.ablate & 2) tk = __builtin_amdgcn_s_memtime();
    if constexpr (LATE) mfma_block(S0, NO, YES);
    else softmax_tile(0);
    for (int j = 1; j < ntiles; j += 2) {
      tile_steps(latec, S1, j);
      if (j + 1 < ntiles) tile_steps(latec, S0, j + 1);
    }
    // after the last tile: its V^T P product only
    wait_and_sync();
    auto last = [&]() __attribute__((always_inline)) {
      if (ntiles & 1) mfma_block(S1, YES, NO);
      else mfma_block(S0, YES, NO);
    };
    if constexpr (LATE) softmax_tile(ntiles - 1);
    else last();
    __syncthreads();
    if constexpr (LATE) last();
  };
  stage(S0, NO, 0);
  if (late) run(YES);
  else run(NO);

  // ---- finish: combine the two half-row sums, normalise, store -------------------
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv_l = 1.0f / l_tot;
  const int qrow = q0 + r;
  const bool q_ok = qrow < p.Sq;
  if (p.lse && q_ok && hh == 0) {
    p.lse[((int64_t)b * p.H + head) * p.Sq + qrow] =
        (m_run * c + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f;
  }
  unsigned short* op = p.o + b * p.o_sb + (int64_t)(q_ok ? qrow : 0) * p.o_ss + head * p.o_sh;
#pragma unroll
  for (int d = 0; d < DVB; ++d) {
    unsigned int pk[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      pk[g][0] = pack_bf2(oacc[d][4 * g] * inv_l, oacc[d][4 * g + 1] * inv_l);
      pk[g][1] = pack_bf2(oacc[d][4 * g + 2] * inv_l, oacc[d][4 * g + 3] * inv_l);
    }
#pragma unroll
    for (int g = 0; g < 4; g += 2) {
#pragma unroll
      for (int w2 = 0; w2 < 2; ++w2) {
        auto sw = __builtin_amdgcn_permlane32_swap(pk[g][w2], pk[g + 1][w2], false, false);
        pk[g][w2] = sw[0];
        pk[g + 1][w2] = sw[1];
      }
      if (q_ok) {
        const int dv8 = d * 32 + 8 * (g + hh);
        *reinterpret_cast<u32x4*>(op + dv8) = u32x4{pk[g][0], pk[g][1], pk[g + 1][0], pk[g + 1][1]};
      }
    }
  }
  if ((p.ablate & 2) && p.lse) {
    // written over the start of lse after every row of workgroup 0 has been stored
    __syncthreads();
    if (lane == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
      p.lse[3 * wave] = (float)t_mfma;
      p.lse[3 * wave + 1] = (float)t_vec;
      p.lse[3 * wave + 2] = (float)t_bar;
    }
  }
}

template <bool F16>
int launch_attn_pingpong(const AttnParams& p, hipStream_t st) {
  constexpr int LDS = 2 * 2 * 64 * 128 * 2;
  static std::atomic<unsigned long long> attr_devices{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long dev_bit = 1ull << (dev & 63);
  auto kfn = attn_fwd_pingpong_kernel<F16>;
  if (!(attr_devices.load(std::memory_order_acquire) & dev_bit)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      mi355_set_error("flash_attn_fwd: cannot reserve %d B of LDS: %s", LDS, hipGetErrorString(e));
      return MI355_EHIP;
    }
    attr_devices.fetch_or(dev_bit, std::memory_order_release);
  }
  dim3 grid((unsigned)((p.Sq + 255) / 256), (unsigned)p.H, (unsigned)p.B), block(512);
  hipLaunchKernelGGL(kfn, grid, block, LDS, st, p);
  MI355_CHECK_LAUNCH("flash_attn_fwd (two-phase)");
  return MI355_OK;
}

// M16: attn_fwd16_kernel (the 16x16x32 MFMA shapes; head size 128) instead of attn_fwd_kernel
template <int D, bool TRREAD, bool F16, int NW = 4, bool M16 = false, bool ASMV = false>
int launch_attn(const AttnParams& p, hipStream_t st) {
  if constexpr (NW == 4) {
    // Eight waves (256 query rows) share each K/V tile once the launch still has a workgroup
    // for every CU: same occupancy (one 8-wave workgroup per CU instead of two 4-wave ones),
    // same results to the bit, half the L2 -> LDS traffic.
    static std::atomic<int> cus{0};
    int n = cus.load(std::memory_order_relaxed);
    if (n == 0) {
      int d = 0;
      (void)hipGetDevice(&d);
      if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n <= 0)
        n = 256;
      cus.store(n, std::memory_order_relaxed);
    }
    if (((p.Sq + 255) / 256) * (int64_t)p.H * p.B >= n) return launch_attn<D, TRREAD, F16, 8, M16, ASMV>(p, st);
  }
  constexpr int LDS = 2 * 2 * 64 * D * 2;
  // the LDS limit is a per-device function attribute: remember which devices have it
  // callers may launch from several host threads: the per-device flags are atomic (a lost
  // update would only repeat the attribute call)
  static std::atomic<unsigned long long> attr_devices{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long dev_bit = 1ull << (dev & 63);
  void (*kfn)(AttnParams);
  if constexpr (M16) kfn = attn_fwd16_kernel<D, F16, NW>;
  else kfn = attn_fwd_kernel<D, TRREAD, F16, NW, ASMV>;
  if (!(attr_devices.load(std::memory_order_acquire) & dev_bit)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      mi355_set_error("flash_attn_fwd: cannot reserve %d B of LDS: %s", LDS, hipGetErrorString(e));
      return MI355_EHIP;
    }
    attr_devices.fetch_or(dev_bit, std::memory_order_release);
  }
  dim3 grid((unsigned)((p.Sq + NW * 32 - 1) / (NW * 32)), (unsigned)p.H, (unsigned)p.B), block(NW * 64);
  hipLaunchKernelGGL(kfn, grid, block, LDS, st, p);
  MI355_CHECK_LAUNCH("flash_attn_fwd");
  return MI355_OK;
}

}  // namespace

static int attn_entry(const void* q, const void* k, const void* v, void* out, float* lse,
                      int64_t B, int64_t Sq, int64_t Sk, int64_t H, int64_t Hk, int64_t D,
                      int64_t q_sb, int64_t q_ss, int64_t q_sh, int64_t k_sb, int64_t k_ss,
                      int64_t k_sh, int64_t v_sb, int64_t v_ss, int64_t v_sh, int64_t o_sb,
                      int64_t o_ss, int64_t o_sh, float softmax_scale, int causal, bool v_f16,
                      void* stream, const float* qw0 = nullptr, const float* qw1 = nullptr,
                      int64_t q_split = 0, float q_eps = 0.f, const float* qcos = nullptr,
                      const float* qsin = nullptr) {
  MI355_REQUIRE(B == 0 || Sq == 0 || (q && k && v && out), "flash_attn_fwd: null pointer");
  if (causal) {
    mi355_set_error("flash_attn_fwd: causal masking is not built (the DiT path is non-causal)");
    return MI355_ENOSUP;
  }
  MI355_REQUIRE(D == 128 || D == 64, "flash_attn_fwd: head size %ld not in {64,128}", (long)D);
  MI355_REQUIRE(B >= 0 && Sq >= 0 && Sk >= 0 && H > 0 && Hk > 0 && H % Hk == 0,
                "flash_attn_fwd: bad shape B=%ld Sq=%ld Sk=%ld H=%ld Hk=%ld", (long)B, (long)Sq,
                (long)Sk, (long)H, (long)Hk);
  MI355_REQUIRE(Sq < (1ll << 31) && Sk < (1ll << 31) && H < 65536 && B < 65536 &&
                    k_ss < (1ll << 24) && v_ss < (1ll << 24),
                "flash_attn_fwd: dimension too large");
  const int64_t strides[] = {q_sb, q_ss, q_sh, k_sb, k_ss, k_sh, v_sb, v_ss, v_sh, o_sb, o_ss, o_sh};
  for (int64_t s : strides)
    MI355_REQUIRE(s % 8 == 0, "flash_attn_fwd: strides must be multiples of 8 elements");
  MI355_REQUIRE(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)v % 16) == 0 &&
                    ((uintptr_t)out % 16) == 0,
                "flash_attn_fwd: q/k/v/out must be 16-byte aligned");
  MI355_REQUIRE((qw0 == nullptr) == (qw1 == nullptr) && (qcos == nullptr) == (qsin == nullptr),
                "flash_attn_fwd: the query norm weights, and cos / sin, come in pairs");
  MI355_REQUIRE(((uintptr_t)qw0 % 16) == 0 && ((uintptr_t)qw1 % 16) == 0 &&
                    ((uintptr_t)qcos % 16) == 0 && ((uintptr_t)qsin % 16) == 0,
                "flash_attn_fwd: query transform tables must be 16-byte aligned");
  if (B == 0 || Sq == 0) return MI355_OK;
  MI355_REQUIRE(Sk > 0, "flash_attn_fwd: empty key sequence");
  AttnParams p;
  p.q = (const unsigned short*)q;
  p.k = (const unsigned short*)k;
  p.v = (const unsigned short*)v;
  p.o = (unsigned short*)out;
  p.lse = lse;
  p.B = (int)B;
  p.Sq = (int)Sq;
  p.Sk = (int)Sk;
  p.H = (int)H;
  p.Hk = (int)Hk;
  p.q_sb = q_sb; p.q_ss = q_ss; p.q_sh = q_sh;
  p.k_sb = k_sb; p.k_ss = k_ss; p.k_sh = k_sh;
  p.v_sb = v_sb; p.v_ss = v_ss; p.v_sh = v_sh;
  p.o_sb = o_sb; p.o_ss = o_ss; p.o_sh = o_sh;
  p.scale_log2 = softmax_scale * 1.4426950408889634f;
  p.qw0 = qw0;
  p.qw1 = qw1;
  p.q_split = (int)(q_split < 0 ? 0 : (q_split > Sq ? Sq : q_split));
  p.q_eps = q_eps;
  p.qcos = qcos;
  p.qsin = qsin;
  {
    const char* ab = getenv("MI355_ATTN_ABLATE");
    p.ablate = ab ? atoi(ab) : 0;
  }
  hipStream_t st = (hipStream_t)stream;
  if (D == 128 && getenv("MI355_ATTN_PINGPONG") != nullptr)
    return v_f16 ? launch_attn_pingpong<true>(p, st) : launch_attn_pingpong<false>(p, st);
  // A/B switch while the 16x16 MFMA form is being measured (read per call)
  {
    const char* m16 = getenv("MI355_ATTN_MFMA16");
    if (D == 128 && m16 && atoi(m16) != 0)
      return v_f16 ? launch_attn<128, true, true, 4, true>(p, st)
                   : launch_attn<128, true, false, 4, true>(p, st);
  }
  // head size 128: the V fragment reads as inline asm (same bits, 1-1.5 % less time;
  // MI355_ATTN_ASMV=0 selects the builtin form for A/B timing; read per call)
  const char* asmv = getenv("MI355_ATTN_ASMV");
  if (D == 128 && !(asmv && atoi(asmv) == 0))
    return v_f16 ? launch_attn<128, true, true, 4, false, true>(p, st)
                 : launch_attn<128, true, false, 4, false, true>(p, st);
  if (v_f16) return D == 128 ? launch_attn<128, true, true>(p, st) : launch_attn<64, true, true>(p, st);
  static const bool no_tr = getenv("MI355_ATTN_NO_TR") != nullptr;  // debugging aid
  if (no_tr)
    return D == 128 ? launch_attn<128, false, false>(p, st) : launch_attn<64, false, false>(p, st);
  return D == 128 ? launch_attn<128, true, false>(p, st) : launch_attn<64, true, false>(p, st);
}

extern "C" int mi355_flash_attn_fwd(const void* q, const void* k, const void* v, void* out,
                                    float* lse, int64_t B, int64_t Sq, int64_t Sk, int64_t H,
                                    int64_t Hk, int64_t D, int64_t q_sb, int64_t q_ss,
                                    int64_t q_sh, int64_t k_sb, int64_t k_ss, int64_t k_sh,
                                    int64_t v_sb, int64_t v_ss, int64_t v_sh, int64_t o_sb,
                                    int64_t o_ss, int64_t o_sh, float softmax_scale, int causal,
                                    void* stream) {
  return attn_entry(q, k, v, out, lse, B, Sq, Sk, H, Hk, D, q_sb, q_ss, q_sh, k_sb, k_ss, k_sh, v_sb,
                    v_ss, v_sh, o_sb, o_ss, o_sh, softmax_scale, causal, false, stream);
}

extern "C" int mi355_flash_attn_fwd_f16v(const void* q, const void* k, const void* v, void* out,
                                         float* lse, int64_t B, int64_t Sq, int64_t Sk, int64_t H,
                                         int64_t Hk, int64_t D, int64_t q_sb, int64_t q_ss,
                                         int64_t q_sh, int64_t k_sb, int64_t k_ss, int64_t k_sh,
                                         int64_t v_sb, int64_t v_ss, int64_t v_sh, int64_t o_sb,
                                         int64_t o_ss, int64_t o_sh, float softmax_scale,
                                         int causal, void* stream) {
  return attn_entry(q, k, v, out, lse, B, Sq, Sk, H, Hk, D, q_sb, q_ss, q_sh, k_sb, k_ss, k_sh, v_sb,
                    v_ss, v_sh, o_sb, o_ss, o_sh, softmax_scale, causal, true, stream);
}

extern "C" int mi355_flash_attn_fwd_ex(const mi355_attn_args* a, void* stream) {
  MI355_REQUIRE(a != nullptr, "flash_attn_fwd_ex: null args");
  return attn_entry(a->q, a->k, a->v, a->out, a->lse, a->B, a->Sq, a->Sk, a->H, a->Hk, a->D, a->q_sb,
                    a->q_ss, a->q_sh, a->k_sb, a->k_ss, a->k_sh, a->v_sb, a->v_ss, a->v_sh, a->o_sb,
                    a->o_ss, a->o_sh, a->softmax_scale, a->causal, a->v_is_f16 != 0, stream,
                    a->q_norm_w0, a->q_norm_w1, a->q_split, a->q_eps, a->q_cos, a->q_sin);
}
