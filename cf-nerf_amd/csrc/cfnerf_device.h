// cfnerf_device.h - device-side building blocks shared by the forward and backward kernels.
// gfx950 only: 64-wide wavefronts, v_mfma_f32_32x32x2_f32 (exact fp32), LDS-resident activation tile.
#pragma once
#include <hip/hip_runtime.h>

#include "cfnerf_layout.h"

namespace cfnerf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kThreads = 256;          // 4 waves per workgroup
constexpr int kWaves = 4;

#define CFN_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// lane id the optimiser cannot see through: address arithmetic derived from it stays at its use
// instead of being hoisted out of the persistent tile loop (which spilt hundreds of registers).
__device__ __forceinline__ int lane_id_opaque() {
    int l = threadIdx.x & 63;
    asm volatile("" : "+v"(l));
    return l;
}
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }

// ---- the kernel arguments as MEMORY, fetched per phase, not as registers held across the persistent tile loop -------------------
// The fused kernels take ~1.2 KB of arguments (the pointer block + the 960-byte operand table).  Read as by-value parameters, every
// field a tile touches - and every predicate / address the optimiser derives from one - is loop-invariant, so it is hoisted in front of
// the persistent loop: ~300 scalars for ~100 SGPRs, the rest spilt to VGPR lanes (v_writelane) and fetched back with ~550 v_readlane
// per (tile, wave) - vector-pipe instructions, which on this chip come straight out of the MFMA issue slots.  Instead a PHASE (a trunk
// layer, the heads, the flows ...) takes a pointer to the kernarg segment that the optimiser cannot see through and s_loads what it
// needs from the constant cache: scalar-unit work under the phase's first MFMAs, nothing live across phases.
#define CFN_KCONST __attribute__((address_space(4)))
template <class KA, bool FRESH = true>
__device__ __forceinline__ const CFN_KCONST KA* kernarg_fresh() {
    const CFN_KCONST KA* p = (const CFN_KCONST KA*)__builtin_amdgcn_kernarg_segment_ptr();
    if (FRESH) asm volatile("" : "+s"(p));
    return p;
}
// one kernel source, either scheme per instantiation: the per-phase view of the kernarg segment or the by-value parameter itself
template <bool MEM> struct karg_pick;
template <> struct karg_pick<true> {
    template <class S> static __device__ __forceinline__ const CFN_KCONST S& get(const CFN_KCONST S& mem, const S&) { return mem; }
};
template <> struct karg_pick<false> {
    template <class S> static __device__ __forceinline__ const S& get(const CFN_KCONST S&, const S& by_value) { return by_value; }
};
// a wave-uniform value (an SGPR or an SGPR pair) the optimiser cannot see through from here on
template <class V>
__device__ __forceinline__ V sgpr_fresh(V v) {
    asm volatile("" : "+s"(v));
    return v;
}
// ... and a device-memory pointer: laundered AS a global-address-space pointer, so what is reached through it stays global_load /
// global_store (a laundered generic pointer has lost its provenance and every access through it becomes a flat_* instruction)
template <class V>
__device__ __forceinline__ V* sgpr_fresh(V* p) {
    typedef V __attribute__((address_space(1)))* gptr;
    gptr g = (gptr)p;
    asm volatile("" : "+s"(g));
    return (V*)g;
}
template <bool FRESH, class V>
__device__ __forceinline__ V sgpr_fresh_if(V v) { return FRESH ? sgpr_fresh(v) : v; }
// "these scalars of a phase are fetched TOGETHER": every argument must be in its register here, so their s_loads are issued back to
// back in front of ONE wait.  Left to itself the compiler puts each load at its value's first use with a full wait in front of that
// use - a dependent chain of constant-cache round trips (~250 cycles each) through the phase.
template <class V>
__device__ __forceinline__ void sgpr_needed(const V& v) { asm volatile("" :: "s"(v)); }
template <class... V>
__device__ __forceinline__ void fetched_together(const V&... v) { (sgpr_needed(v), ...); }
// by-value copy of a table entry out of the kernarg segment (scalar loads)
template <class V>
__device__ __forceinline__ V kload(const CFN_KCONST V& src) {
    V r;
    __builtin_memcpy(&r, &src, sizeof(V));
    return r;
}
template <class V>
__device__ __forceinline__ V kload(const V& src) { return src; }
// an operand-table entry is 16 bytes: ONE s_load_dwordx4 (field by field the compiler issues a dword load + a full wait per field, at
// each field's first use)
typedef uint32_t u32x4_k __attribute__((ext_vector_type(4), aligned(4)));
template <>
__device__ __forceinline__ SubL kload<SubL>(const CFN_KCONST SubL& src) {
    const u32x4_k v = *reinterpret_cast<const CFN_KCONST u32x4_k*>(&src);
    SubL r;
    __builtin_memcpy(&r, &v, sizeof(SubL));
    return r;
}

// activation row stride (floats): +4 keeps ds_read_b128 of 16 consecutive rows conflict-free
// (row stride = 16 B mod 256 B) and rows 16-B aligned.  >= 128 so the theta tile fits.
__host__ __device__ constexpr int act_ld(int W) { return (W > kThetaAll ? W : kThetaAll) + 4; }

// accumulators start at the bias of the lane's column: the epilogue then needs no add (a lane's 32 fragment values of a
// tile all belong to one output column)
template <int NTW>
__device__ __forceinline__ void acc_init(f32x16 (&acc)[2][NTW], const float (&bias)[NTW]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = bias[j];
}

template <int NTW>
__device__ __forceinline__ void acc_zero(f32x16 (&acc)[2][NTW]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

// 16*NV MFMAs of one k-chunk (8 k values): a0/a1 = A fragments of the two row tiles, b[j] = B fragments
// wave priorities: low inside the MFMA loops, raised in the epilogues (+1 % over none; the inverse pair +0.5 %: profiles/EXPERIMENTS.md)
#define CFN_SETPRIO(x) __builtin_amdgcn_s_setprio(x)
#define CFN_MMA_PRIO 0
#define CFN_EPI_PRIO 2
template <int NTW, int NV>
__device__ __forceinline__ void mma_block(f32x16 (&acc)[2][NTW], const f32x4 a0, const f32x4 a1, const f32x4 (&b)[NTW]) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            acc[0][j] = CFN_MFMA(a0[c], b[j][c], acc[0][j]);
            acc[1][j] = CFN_MFMA(a1[c], b[j][c], acc[1][j]);
        }
}

// k-loop, unrolled by two with ping-pong operand registers: the loads of chunk k+1 are issued
// before the MFMAs of chunk k, so L2 / LDS latency hides under a full block of MFMAs and no
// register copies are needed.  The packed operand is read through ONE buffer descriptor over the packed-weight array: a
// lane keeps a single byte offset (16 * lane) for every n-tile and k-chunk, what differs is a SCALAR offset (so[j] + 1 KB per
// chunk) - no 64-bit vector address arithmetic between the MFMAs (it came out of their issue slots: 4 of the loop's 6 VALU).
template <int NTW, int NV, bool PRE = false>
__device__ __forceinline__ void mma_loop(f32x16 (&acc)[2][NTW], const __amdgpu_buffer_rsrc_t wr, const int voff, const int (&so)[NTW],
                                         const float* a_ptr, int lda, int KC, const f32x4* bpre = nullptr) {
    f32x4 bA[NTW], bB[NTW], a0A, a1A, a0B, a1B;
    auto bload = [&](int j, int kc) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, voff, so[j] + kc * 1024, 0)); };
    CFN_SETPRIO(CFN_MMA_PRIO);
#pragma unroll
    for (int j = 0; j < NV; ++j) bA[j] = PRE ? bpre[j] : bload(j, 0);      // PRE: chunk 0 was fetched ahead, under the previous layer's epilogue
    a0A = *reinterpret_cast<const f32x4*>(a_ptr);
    a1A = *reinterpret_cast<const f32x4*>(a_ptr + 32 * lda);
    int kc = 0;
    for (; kc + 1 < KC; kc += 2) {
#pragma unroll
        for (int j = 0; j < NV; ++j) bB[j] = bload(j, kc + 1);
        a0B = *reinterpret_cast<const f32x4*>(a_ptr + (kc + 1) * 8);
        a1B = *reinterpret_cast<const f32x4*>(a_ptr + 32 * lda + (kc + 1) * 8);
        mma_block<NTW, NV>(acc, a0A, a1A, bA);
        const int k2 = (kc + 2 < KC) ? kc + 2 : KC - 1;       // clamped: the last prefetch re-reads valid data
#pragma unroll
        for (int j = 0; j < NV; ++j) bA[j] = bload(j, k2);
        a0A = *reinterpret_cast<const f32x4*>(a_ptr + k2 * 8);
        a1A = *reinterpret_cast<const f32x4*>(a_ptr + 32 * lda + k2 * 8);
        mma_block<NTW, NV>(acc, a0B, a1B, bB);
    }
    if (kc < KC) mma_block<NTW, NV>(acc, a0A, a1A, bA);       // odd KC tail (operands already loaded)
    CFN_SETPRIO(CFN_EPI_PRIO);
}

// acc[i][j] += A[rows i*32..+31][0..8*kc) * B(tile nt0 + j*nts)       (one GEMM segment)
//   A: LDS, row-major, stride lda floats, 64 rows.   B: packed operand `s` in global (L2-resident).
// Lane l holds A[row l&31][k0 + 4*(l>>5) + c] and B[k0 + 4*(l>>5) + c][col l&31], c = 0..3:
// one ds_read_b128 + one global_load_dwordx4 feed four MFMAs per tile.
template <int NTW>
__device__ __forceinline__ void mma_seg(f32x16 (&acc)[2][NTW], const SubL s, int nt0, int nts,
                                        const float* __restrict__ wp, const float* lds_a, int lda) {
    const int lane = lane_id_opaque();
    const int KC = s.kc;
    int nvalid = 0;
#pragma unroll
    for (int j = 0; j < NTW; ++j) nvalid += (nt0 + j * nts < (int)s.nt) ? 1 : 0;
    if (nvalid == 0) return;
    const float* a_ptr = lds_a + (lane & 31) * lda + 4 * (lane >> 5);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp), 0, 0x7ffffff0, 0x00020000);
    int so[NTW];                                // byte offset of fragment 0 of the wave's n-tiles: wave-uniform
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        int nt = nt0 + j * nts;
        if (nt >= (int)s.nt) nt = nt0;          // never dereferenced (j >= nvalid)
        so[j] = __builtin_amdgcn_readfirstlane((int)((s.w_off + (unsigned)nt * KC * 256u) * 4u));
    }
    // Every operand of the supported widths (64 / 128 / 256 / 512, heads of 32 / 64 / 96 columns) gives a wave either all
    // NTW of its n-tiles or none (NT is a multiple of the wave count or smaller than it): ONE loop variant per call site, so the
    // accumulators stay in place (a set of partial variants made the compiler shuffle 32 accumulator registers through
    // v_mov_b64 around every GEMM segment).
    mma_loop<NTW, NTW>(acc, wr, lane * 16, so, a_ptr, lda, KC);
}

// The first B fragments (k-chunk 0) of a dense block, fetched AHEAD of it: issued before the barrier / epilogue that precedes the
// block, they cross L2 while that runs, and mma_seg_pre starts its first MFMAs without waiting for them.
template <int NTW>
__device__ __forceinline__ void b_prefetch(const SubL s, int nt0, int nts, const float* __restrict__ wp, f32x4 (&bpre)[NTW]) {
    const int lane = lane_id_opaque();
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp), 0, 0x7ffffff0, 0x00020000);
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        int nt = nt0 + j * nts;
        if (nt >= (int)s.nt) nt = nt0 < (int)s.nt ? nt0 : 0;
        const int so = __builtin_amdgcn_readfirstlane((int)((s.w_off + (unsigned)nt * s.kc * 256u) * 4u));
        bpre[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, lane * 16, so, 0));
    }
}
template <int NTW>
__device__ __forceinline__ void mma_seg_pre(f32x16 (&acc)[2][NTW], const SubL s, int nt0, int nts, const float* __restrict__ wp,
                                            const float* lds_a, int lda, const f32x4 (&bpre)[NTW]) {
    const int lane = lane_id_opaque();
    const int KC = s.kc;
    int nvalid = 0;
#pragma unroll
    for (int j = 0; j < NTW; ++j) nvalid += (nt0 + j * nts < (int)s.nt) ? 1 : 0;
    if (nvalid == 0) return;
    const float* a_ptr = lds_a + (lane & 31) * lda + 4 * (lane >> 5);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp), 0, 0x7ffffff0, 0x00020000);
    int so[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        int nt = nt0 + j * nts;
        if (nt >= (int)s.nt) nt = nt0;
        so[j] = __builtin_amdgcn_readfirstlane((int)((s.w_off + (unsigned)nt * KC * 256u) * 4u));
    }
    mma_loop<NTW, NTW, true>(acc, wr, lane * 16, so, a_ptr, lda, KC, bpre);
}

// ================= opt-in split-bf16 ("bf16x3") operand path =====================================
// An fp32 value v is carried as hi = bf16(v), lo = bf16(v - hi); a product a*b is evaluated as
// a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (the dropped
// a_lo*b_lo term is ~2^-18 relative).  LDS activation words hold {hi (low half), lo (high half)}.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
enum { PREC_F32 = 0, PREC_BF16X3 = 1 };

__device__ __forceinline__ unsigned pack_hl(float v) {
    const __bf16 h = (__bf16)v;
    const __bf16 l = (__bf16)(v - (float)h);
    return (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
}
// LDS activation tiles.  fp32 mode: row-major floats, row stride `ld`.  Split-bf16 mode: the SAME row stride in bytes (4 * ld), a
// row holding its hi plane (bf16 columns from byte 0) and its lo plane (from byte 2 ld + 8: 16-byte aligned, ld = width + 4, so the
// two planes fill the 4 ld bytes exactly): the 8 consecutive k a lane
// feeds to v_mfma_f32_32x32x16_bf16 are then 16 contiguous bytes per plane - one ds_read_b128 each, no de-interleave - and the
// row stride keeps the conflict-free bank pattern of the fp32 tile.  An element is addressed as (row pointer, ld, column).
template <int PREC>
__device__ __forceinline__ void act_store(float* row, int ld, int col, float v) {
    if (PREC == PREC_BF16X3) {
        __bf16* r = reinterpret_cast<__bf16*>(row);
        const __bf16 h = (__bf16)v;
        r[col] = h;
        r[ld + 4 + col] = (__bf16)(v - (float)h);
    } else {
        row[col] = v;
    }
}
// two elements of the same column in two rows: in the split-bf16 mode one v_cvt_pk_bf16_f32 converts both hi parts and one
// both lo parts (3 vector instructions per element instead of 4), the halves of a pair go out as ds_write_b16 / _d16_hi
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <int PREC>
__device__ __forceinline__ void act_store2(float* row_a, float* row_b, int ld, int col, float va, float vb) {
    if (PREC == PREC_BF16X3) {
        f32x2_t v; v[0] = va; v[1] = vb;
        const bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
        const unsigned hu = __builtin_bit_cast(unsigned, h);
        f32x2_t r; r[0] = va - __uint_as_float(hu << 16); r[1] = vb - __uint_as_float(hu & 0xffff0000u);
        const bf16x2_t l = __builtin_convertvector(r, bf16x2_t);
        __bf16* pa = reinterpret_cast<__bf16*>(row_a);
        __bf16* pb = reinterpret_cast<__bf16*>(row_b);
        pa[col] = h[0]; pb[col] = h[1];
        pa[ld + 4 + col] = l[0]; pb[ld + 4 + col] = l[1];
    } else {
        row_a[col] = va; row_b[col] = vb;
    }
}
template <int PREC>
__device__ __forceinline__ float act_load(const float* row, int ld, int col) {
    if (PREC == PREC_BF16X3) {
        const __bf16* r = reinterpret_cast<const __bf16*>(row);
        return (float)r[col] + (float)r[ld + 4 + col];
    }
    return row[col];
}
#define CFN_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// acc[i][j] += A * B for one operand in the split-bf16 format.  Per 16-k chunk and tile: three MFMAs.
// A chunk is only 12*NV MFMAs (~400-800 cycles) - less than the L2 latency - so the B fragments (global) run
// kPre16 chunks ahead in a rotating register ring; the A fragments (LDS, short latency) one chunk ahead.
// a_row: this lane's row of the A tile (row lane & 31 of row-tile 0), a_col: bf16 column of its first k (col0 + 8 * (lane >> 5)).
template <int NTW, int NV, int kPre16>
__device__ __forceinline__ void mma_loop16(f32x16 (&acc)[2][NTW], const __amdgpu_buffer_rsrc_t wr, const int voff, const int (&so)[NTW],
                                           const float* a_row, int a_col, int lda, int KC) {
    bf16x8 rb[kPre16][NTW][2];    // [ring slot][n tile][plane]
    bf16x8 ra[2][2][2];           // [buf][row tile][plane]
    auto issue_b = [&](int slot, int kc) {      // one descriptor, one lane offset, scalar chunk / plane offsets (see mma_loop)
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            rb[slot][j][0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wr, voff, so[j] + kc * 2048, 0));
            rb[slot][j][1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wr, voff, so[j] + kc * 2048 + 1024, 0));
        }
    };
    auto issue_a = [&](int buf, int kc) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const __bf16* q = reinterpret_cast<const __bf16*>(a_row + i * 32 * lda) + a_col + kc * 16;
            ra[buf][i][0] = *reinterpret_cast<const bf16x8*>(q);              // hi plane
            ra[buf][i][1] = *reinterpret_cast<const bf16x8*>(q + lda + 4);    // lo plane (2 lda + 8 bytes further)
        }
    };
    auto compute = [&](int buf, int slot) {
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc[i][j] = CFN_MFMA16(ra[buf][i][1], rb[slot][j][0], acc[i][j]);      // small terms first: lo * hi
                acc[i][j] = CFN_MFMA16(ra[buf][i][0], rb[slot][j][1], acc[i][j]);      // hi * lo
                acc[i][j] = CFN_MFMA16(ra[buf][i][0], rb[slot][j][0], acc[i][j]);      // hi * hi
            }
    };
    const int last = KC - 1;
    CFN_SETPRIO(CFN_MMA_PRIO);
#pragma unroll
    for (int q = 0; q < kPre16; ++q) issue_b(q, min(q, last));
    issue_a(0, 0);
    // unrolled by 6 (a multiple of the ring depth 2 or 3 and of the A ping-pong) so every register index is static
    for (int kc = 0; kc < KC; kc += 6) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int k = kc + u;
            if (k < KC) {
                issue_a((u + 1) & 1, min(k + 1, last));
                compute(u & 1, u % kPre16);
                issue_b(u % kPre16, min(k + kPre16, last));
            }
        }
    }
    CFN_SETPRIO(CFN_EPI_PRIO);
}

template <int NTW, int PRE>
__device__ __forceinline__ void mma_seg16(f32x16 (&acc)[2][NTW], const SubL s, int nt0, int nts, const __bf16* __restrict__ wp16,
                                          const float* lds_a, int lda, int col0) {
    const int lane = lane_id_opaque();
    const int KC = s.kc16();
    int nvalid = 0;
#pragma unroll
    for (int j = 0; j < NTW; ++j) nvalid += (nt0 + j * nts < (int)s.nt) ? 1 : 0;
    if (nvalid == 0) return;
    const float* a_row = lds_a + (lane & 31) * lda;
    const int a_col = col0 + 8 * (lane >> 5);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(wp16), 0, 0x7ffffff0, 0x00020000);
    int so[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        int nt = nt0 + j * nts;
        if (nt >= (int)s.nt) nt = nt0;
        so[j] = __builtin_amdgcn_readfirstlane((int)(s.w16_off * 2u + (unsigned)nt * KC * 2048u));
    }
    mma_loop16<NTW, NTW, PRE>(acc, wr, lane * 16, so, a_row, a_col, lda, KC);        // all or none of a wave's n-tiles exist: see mma_seg
}

// precision-dispatching wrapper used by the fused kernels
// PRE: depth of the B-fragment ring of the bf16 loop (2 when two workgroups share a CU, 3 when a wave is alone on its SIMD)
// lds_a: the A tile (row 0), lda: its row stride in floats, col0: first column of the operand inside the tile
template <int NTW, int PREC, int PRE = 2>
__device__ __forceinline__ void mma_any(f32x16 (&acc)[2][NTW], const SubL s, int nt0, int nts, const float* __restrict__ wp,
                                        const __bf16* __restrict__ wp16, const float* lds_a, int lda, int col0 = 0) {
    if (PREC == PREC_BF16X3) mma_seg16<NTW, PRE>(acc, s, nt0, nts, wp16, lds_a, lda, col0);
    else mma_seg<NTW>(acc, s, nt0, nts, wp, lds_a + col0, lda);
}

// streaming (touch-once) global traffic: keep it from evicting the L2-resident packed weights
// A narrow head (1 or 2 n-tiles) would keep one or two waves busy for the whole K: split K over the waves instead.
// Wave w takes n-tile w % nt and k-part w / nt of (n_waves / nt) parts; the partial tiles are summed by the caller.
template <int PREC, int PRE>
__device__ __forceinline__ void mma_ksplit(f32x16 (&acc)[2][1], const SubL s, int wave, int n_waves, const float* __restrict__ wp,
                                           const __bf16* __restrict__ wp16, const float* lds_a, int lda) {
    const int lane = lane_id_opaque();
    // parts of ceil(KC / nparts) chunks: the last part is shorter (or empty: that wave contributes its zeros) when the chunk count
    // is not a multiple of the part count (netwidth 320 / 448 in the split-bf16 mode).  No early exit: an empty part runs the loops
    // with a zero trip count (their prefetches read in-range or bounds-checked addresses and feed no MFMA).
    const int ntc = (int)s.nt, nt = wave % ntc, part = wave / ntc, nparts = n_waves / ntc;
    if (PREC == PREC_BF16X3) {
        const int KC = s.kc16(), kcp = (KC + nparts - 1) / nparts, k0 = min(part * kcp, KC), cnt = min(kcp, KC - k0);
        const float* a_row = lds_a + (lane & 31) * lda;
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(wp16), 0, 0x7ffffff0, 0x00020000);
        const int so[1] = {__builtin_amdgcn_readfirstlane((int)(s.w16_off * 2u + ((unsigned)nt * KC + k0) * 2048u))};
        mma_loop16<1, 1, PRE>(acc, wr, lane * 16, so, a_row, 8 * (lane >> 5) + k0 * 16, lda, cnt);
    } else {
        const int KC = s.kc, kcp = (KC + nparts - 1) / nparts, k0 = min(part * kcp, KC), cnt = min(kcp, KC - k0);
        const float* a_ptr = lds_a + (lane & 31) * lda + 4 * (lane >> 5) + k0 * 8;
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp), 0, 0x7ffffff0, 0x00020000);
        const int so[1] = {__builtin_amdgcn_readfirstlane((int)((s.w_off + ((unsigned)nt * KC + k0) * 256u) * 4u))};
        mma_loop<1, 1>(acc, wr, lane * 16, so, a_ptr, lda, cnt);
    }
}

__device__ __forceinline__ void st_stream(float* p, float v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ float ld_stream(const float* p) { return __builtin_nontemporal_load(p); }

// C/D fragment of v_mfma_f32_32x32x2_f32: lane l, register r -> row (r&3) + 8*(r>>2) + 4*(l>>5), col l&31
__device__ __forceinline__ int frag_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

enum { ACT_NONE = 0, ACT_RELU = 1 };

// Write acc (+bias, activation) to the LDS tile (row-major, stride ld, column offset col0) and
// optionally to a row-major global stash (stride gld) for rows < rows_valid.
// bias of this lane's column in each of the wave's n-tiles, fetched BEFORE the MFMA loop so its L2 latency hides under it
template <int NTW>
__device__ __forceinline__ void load_bias(const SubL s, int nt0, int nts, const float* __restrict__ wp, float (&bv)[NTW]) {
    const int lane = lane_id_opaque();
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int nt = nt0 + j * nts;
        bv[j] = (nt < (int)s.nt && s.b_off != 0xffffffffu) ? wp[s.b_off + nt * 32 + (lane & 31)] : 0.f;
    }
}

// ReLU as ONE integer max on the bit pattern (negative floats and -0 are negative ints -> +0): fmaxf costs two v_max
// (the compiler first quiets a possible signalling NaN of the MFMA result with max(x, x))
__device__ __forceinline__ float relu_f(float v) { return __int_as_float(max(__float_as_int(v), 0)); }

// A row-major [rows x ld] slab of a touch-once HBM stream (activation stash, pre-activation gradients) as a buffer
// descriptor that covers exactly its valid rows: a lane keeps ONE byte offset (its first row and column), the row of
// each fragment element is a SCALAR offset (rr * ld * 4: SALU, not VALU - on this chip every VALU instruction comes
// out of the MFMA issue slots), and rows past the ragged end of the last tile fall outside the descriptor and are
// dropped by the hardware bounds check, so there is no per-element exec-mask code either.  Non-temporal: the stream
// must not evict the L2-resident packed weights.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t slab_rsrc(float* base, int rows_valid, int ld) {
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, rows_valid * ld * 4, 0x00020000);
}
__device__ __forceinline__ void slab_store(__amdgpu_buffer_rsrc_t r, int voff_bytes, int soff_bytes, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff_bytes, soff_bytes, /*nt*/ 2);
}

// Copy-out of a tile the epilogue has just written to LDS (row-major, stride ld) to its row-major HBM slab, as 16-byte pieces:
// 64 lanes x 16 B = 1 KB contiguous per instruction (one whole row of a 256-wide stream).  Storing the fragment elements from the
// accumulator registers instead costs one buffer_store_dword per element - 256 B per instruction in two 128-B pieces - and the
// texture-address unit, not HBM, was what the 1.5 GB of activation stash cost the train forward (~65 us per launch):
// four times fewer vector-memory instructions move the same bytes.  No VALU work: ds_read_b128 + buffer_store_dwordx4, issued
// right after the barrier that follows the epilogue, in front of the next MFMA block (which only READS the tile; whatever WRITES the
// tile next sits behind another barrier, which a wave only reaches after its own copy-out reads have returned).  Rows past a
// ragged tile fall outside the descriptor.  fp32 tiles only (the split-bf16 tiles hold hi / lo planes).  Used by the train forward
// (-27 us at C2, -25 us at W = 512).  Backward-data keeps its element stores: the same scheme there measured -4 us at W = 256 and
// +18 us at W = 512, and its last copy-out of a tile would need one more barrier in front of the next tile's first LDS write.
constexpr bool kStashFromLds = true;
template <int WIDTH, int NTHR>
__device__ __forceinline__ void stash_rows(const float* lds, int ld, float* __restrict__ gdst, int rows_valid) {
    constexpr int QPR = WIDTH / 4, TOTAL = kTileM * QPR;              // 16-byte pieces per row / per tile
    static_assert(TOTAL % NTHR == 0, "tile pieces must divide over the workgroup");
    const __amdgpu_buffer_rsrc_t sink = slab_rsrc(gdst, rows_valid, WIDTH);
    const int tid = threadIdx.x;
    // batches of 4 pieces (16 registers in flight, not 64: the accumulators are dead here but the kernel has no registers to spare)
    constexpr int PER = TOTAL / NTHR, B = PER % 4 == 0 ? 4 : (PER % 2 == 0 ? 2 : 1);
#pragma unroll 1
    for (int b = 0; b < PER / B; ++b) {
        u32x4 v[B];
#pragma unroll
        for (int i = 0; i < B; ++i) {
            const int idx = tid + (b * B + i) * NTHR, row = idx / QPR, q = idx - row * QPR;
            v[i] = *reinterpret_cast<const u32x4*>(lds + row * ld + 4 * q);
        }
#pragma unroll
        for (int i = 0; i < B; ++i) {
            const int idx = tid + (b * B + i) * NTHR, row = idx / QPR, q = idx - row * QPR;
            __builtin_amdgcn_raw_buffer_store_b128(v[i], sink, (row * WIDTH + 4 * q) * 4, 0, /*nt*/ 2);
        }
    }
}

// ReLU mask of a lane's 32-row output fragment as ONE word, element e = i * 16 + r at bit 31 - e: built with one
// v_sub + one v_alignbit per element (the sign bit of 0 - bits(v) is set iff v > 0 for a post-ReLU v), read back in
// the backward with one v_bfe_i32 (sign-extended 1-bit field = all-ones / zero mask).
__device__ __forceinline__ uint32_t relu_bit_push(uint32_t bits, float v_post_relu) {
    return __builtin_amdgcn_alignbit(bits, (uint32_t)(0 - __float_as_int(v_post_relu)), 31);
}
__device__ __forceinline__ float relu_bit_apply(uint32_t bits, int e, float v) {
    const int m = __builtin_amdgcn_sbfe((int)bits, 31 - e, 1);
    return __int_as_float(__float_as_int(v) & m);
}

// ---- the Q4 layout of a wide stash stream ("fragment quads", round 5).  An MFMA output fragment gives lane l (column l & 31, half
// h = l >> 5) registers r = 4 g + e: rows e + 8 g + 4 h of its column - four CONSECUTIVE ROWS per group g, i.e. four floats that are
// 1 KB apart in a row-major [P, 256] array.  Row-major the stash therefore costs a dword store per element (or a round trip through
// LDS, stash_rows); a timing build that wrote the same bytes as the fragments ARE - one buffer_store_dwordx4 per group, 64 lanes x 16 B
// = one contiguous 1-KB piece per instruction, straight from the accumulator registers - measured -1.3 % on the train forward and
// -3.5 % on backward-data (profiles/EXPERIMENTS.md).  So the wide streams of the train step (h, feature, v and their pre-activation
// gradients) - in practice the trunk streams h[l], g_h[l] and g_feat, see fused_fwd_kernel - are laid out that way whenever the batch's tiles are whole (S a multiple of 64; fp32 mode):
//   tile t = 64 consecutive points, half i = 32 points, n-tile nt = 32 columns, group g, lane, e:
//   float offset  t * 64 C + ((i * C/32 + nt) * 4 + g) * 256 + lane * 4 + e    holds   A[64 t + 32 i + 8 g + 4 (lane >> 5) + e][32 nt + (lane & 31)]
// A tile occupies the same 64 C floats as row-major, a 32-point half-tile is contiguous (one LDS stage of the weight-gradient
// kernels), and those kernels consume a piece as it is: ONE ds_read_b128 per lane feeds four MFMA k-steps of a 32-column tile
// (A and B use the same point assignment 8 g + 4 h + e, so the contraction order is consistent).
__device__ __forceinline__ int q4_piece(int i, int nt, int g, int n_tiles) { return ((i * n_tiles + nt) * 4 + g) * 1024; }      // byte offset of a 1-KB piece inside its tile
// One lane's 16 bytes of a piece.  The piece offset rides in the VECTOR offset, soffset is the literal 0 - on purpose: with the offset in
// an SGPR (the form every other stream uses) a 16-byte buffer store reads its data registers LATE on gfx950, and the epilogue's next
// v_and / v_max into the same four VGPRs - a few instructions on - reached memory instead: sparse, run-to-run different garbage in g_h
// under two workgroups per CU (round 5, tests/tools/determinism_diag.py: lanes 12-15 of every row of 16, first dword, values of the NEXT
// group).  LLVM's hazard recognizer knows the ">64-bit store, then VALU write of its data" hazard only for the form WITHOUT an SGPR offset
// (GCNHazardRecognizer::createsVALUHazard) and inserts the wait state there; with the literal 0 the same code is bit-reproducible.
__device__ __forceinline__ void q4_store(u32x4 v, __amdgpu_buffer_rsrc_t sink, int lane, int piece_bytes) {
    __builtin_amdgcn_raw_buffer_store_b128(v, sink, lane * 16 + piece_bytes, 0, /*nt*/ 2);
}

// epilogue of a layer: bias + activation -> LDS tile (and, in the train variants, the activation stash in HBM).
// STASH is 0 (no stash code at all: the eval variants), 1 (row-major stash through a slab descriptor) or 2 (Q4: one dwordx4 per
// group from the registers); store_tiles picks it once per call, so the per-element work is max, ds_write (+ a store, + two ops for the mask bit).
template <int NTW, int ACT, int PREC, bool WANT_BITS, int STASH, bool BIAS_IN_ACC>
__device__ __forceinline__ void store_tiles_impl(const f32x16 (&acc)[2][NTW], const SubL s, int nt0, int nts,
                                                 const float* __restrict__ wp, float* lds_dst, int ld, int col0,
                                                 float* __restrict__ gdst, int gld, int rows_valid,
                                                 uint32_t* __restrict__ mbits, const float* bias_pre) {
    const int lane = lane_id_opaque();
    const int rbase = 4 * (lane >> 5);
    // (STASH == 2 without a stash - a train forward under no_grad: gdst is null, the descriptor is empty and the hardware drops the stores)
    const __amdgpu_buffer_rsrc_t sink = slab_rsrc(gdst, STASH == 1 ? rows_valid : (STASH == 2 && gdst != nullptr) ? kTileM : 0, gld);      // unused (dead code) when STASH == 0
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int nt = nt0 + j * nts;
        if (nt >= (int)s.nt) continue;
        const int col = nt * 32 + (lane & 31);
        const float bv = BIAS_IN_ACC ? 0.f : bias_pre ? bias_pre[j] : (s.b_off != 0xffffffffu) ? wp[s.b_off + col] : 0.f;
        float* lrow = lds_dst + rbase * ld;
        const int lcol = col0 + col;
        const int voff = (rbase * gld + col) * 4;
        uint32_t bits = 0;
        [[maybe_unused]] u32x4 q4v;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {                         // elements r, r + 1: two consecutive rows of this column
                const int rr = i * 32 + (r & 3) + 8 * (r >> 2);       // row = rr + rbase
                float v0 = BIAS_IN_ACC ? acc[i][j][r] : acc[i][j][r] + bv;
                float v1 = BIAS_IN_ACC ? acc[i][j][r + 1] : acc[i][j][r + 1] + bv;
                if (ACT == ACT_RELU) { v0 = relu_f(v0); v1 = relu_f(v1); }
                act_store2<PREC>(lrow + rr * ld, lrow + (rr + 1) * ld, ld, lcol, v0, v1);
                if (STASH == 1) { slab_store(sink, voff, rr * gld * 4, v0); slab_store(sink, voff, (rr + 1) * gld * 4, v1); }
                if (STASH == 2) {                                     // Q4: the four rows of group g = r >> 2 leave as ONE 16-byte store
                    if ((r & 3) == 0) { q4v[0] = __float_as_uint(v0); q4v[1] = __float_as_uint(v1); }
                    else {
                        q4v[2] = __float_as_uint(v0); q4v[3] = __float_as_uint(v1);
                        q4_store(q4v, sink, lane, q4_piece(i, nt, r >> 2, gld >> 5));
                    }
                }
                if (WANT_BITS) { bits = relu_bit_push(bits, v0); bits = relu_bit_push(bits, v1); }
            }
        // ReLU mask of this lane's fragment (32 rows of one column) as one word, in exactly the layout the
        // backward-data kernel's output fragment has: it replaces 32 float loads per lane there
        if (WANT_BITS && mbits != nullptr) mbits[nt * 64 + lane] = bits;
    }
}

// The Q4 stash of a layer WITHOUT the LDS write: acc (+ bias) leaves as pieces straight from the registers, the tile in LDS is written later
// by a plain store_tiles (STASH 0) of the same accumulators.  For the feature head of the train forward: its inputs (the last trunk activations)
// must stay in the LDS tile until the h_alpha head has read them too, and with BOTH heads' accumulators live the Q4 stores of the combined
// epilogue spilt 13 - 37 VGPRs (rounds 5 - 6); issued right after the feature head's own MFMA loop - before the h_alpha accumulators exist -
// they cost no register.  The bias is in the accumulators already (acc_init), there is no activation: the same values as store_tiles<..., Q4>.
template <int NTW>
__device__ __forceinline__ void stash_tiles_q4(const f32x16 (&acc)[2][NTW], const SubL s, int nt0, int nts, float* __restrict__ gdst, int gld) {
    const int lane = lane_id_opaque();
    const __amdgpu_buffer_rsrc_t sink = slab_rsrc(gdst, gdst != nullptr ? kTileM : 0, gld);      // (no stash: empty descriptor, the stores are dropped)
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int nt = nt0 + j * nts;
        if (nt >= (int)s.nt) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = __float_as_uint(acc[i][j][4 * g + e]);
                q4_store(v, sink, lane, q4_piece(i, nt, g, gld >> 5));
            }
    }
}

// MAY_STASH: the row-major element stores may be needed (a stream that is never copied out by stash_rows); Q4: the stream takes the Q4
// layout (a compile-time property of the kernel variant: two epilogue flavours inside ONE kernel spilt 20-30 VGPRs of the train forward) -
// gdst is then the TILE's base, the same address as its row-major slab.
template <int NTW, int ACT, int PREC = PREC_F32, bool WANT_BITS = false, bool MAY_STASH = true, bool BIAS_IN_ACC = false, bool Q4 = false>
__device__ __forceinline__ void store_tiles(const f32x16 (&acc)[2][NTW], const SubL s, int nt0, int nts,
                                            const float* __restrict__ wp, float* lds_dst, int ld, int col0,
                                            float* __restrict__ gdst, int gld, int rows_valid,
                                            uint32_t* __restrict__ mbits = nullptr, const float* bias_pre = nullptr) {
    if (Q4)
        store_tiles_impl<NTW, ACT, PREC, WANT_BITS, 2, BIAS_IN_ACC>(acc, s, nt0, nts, wp, lds_dst, ld, col0, gdst, gld, rows_valid, mbits, bias_pre);
    else if (MAY_STASH && gdst != nullptr)
        store_tiles_impl<NTW, ACT, PREC, WANT_BITS, 1, BIAS_IN_ACC>(acc, s, nt0, nts, wp, lds_dst, ld, col0, gdst, gld, rows_valid, mbits, bias_pre);
    else
        store_tiles_impl<NTW, ACT, PREC, WANT_BITS, 0, BIAS_IN_ACC>(acc, s, nt0, nts, wp, lds_dst, ld, col0, gdst, gld, rows_valid, mbits, bias_pre);
}

// ---- elementwise numerics: the same definitions torch uses on the reference path ---------------
__device__ __forceinline__ float softplus_f(float x) {        // F.softplus(beta=1, threshold=20)
    return x > 20.f ? x : log1pf(expf(x));
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// The flows + composite of one (point, latent) cost ~1300 vector instructions with the libm routines above (16 tanhf at ~27,
// 16 logf in train, 4 softplus, 3 sigmoids with an IEEE division, 1 expf) and this chip issues nothing else while they run: 5 % of a
// tile at K = 16, 10 % at K = 32, 20 % at the reference's default K = 64.  FAST = the same functions on the hardware
// transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp each): tanh = 1 - 2 / (1 + e^2x) (absolute error ~2e-7), softplus =
// ln(1 + e^x) with torch's threshold, sigmoid = 1 / (1 + e^-x), ln = log2 * ln 2: ~250 instructions.  Measured on `raw` / rgb_map /
// entropy against the reference fixtures and the oracle: inside the SAME 1e-5 / 1e-4 bounds (the tests are the gate).  The fused
// kernels take it for K >= kFastFlowsK only: the reference's plumbing and headline configurations (K = 1, 4, 8) keep the libm path
// bit for bit; cfnerf_model_set_flow_math selects either path for every K.
constexpr int kFastFlowsK = 16;
template <bool FAST> struct Num {
    static __device__ __forceinline__ float exp(float x) { return FAST ? __builtin_amdgcn_exp2f(x * 1.4426950408889634f) : expf(x); }
    static __device__ __forceinline__ float ln(float x) { return FAST ? __builtin_amdgcn_logf(x) * 0.6931471805599453f : logf(x); }
    static __device__ __forceinline__ float tanh(float x) {
        return FAST ? __builtin_fmaf(-2.f, __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * 2.8853900817779268f)), 1.f) : tanhf(x);   // (fma: the same number as 1 - 2 r, doubling is exact)
    }
    static __device__ __forceinline__ float sigmoid(float x) {
        return FAST ? __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f)) : sigmoid_f(x);
    }
    static __device__ __forceinline__ float softplus(float x) {
        if (!FAST) return softplus_f(x);
        return x > 20.f ? x : __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(x * 1.4426950408889634f)) * 0.6931471805599453f;
    }
};

// Num<true> on TWO latent samples per lane (the flow phase from 16 latents on, cfnerf_fwd.hip): the multiplies and adds around the hardware
// transcendentals become packed fp32 instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: two lanes' worth per issue slot), the
// transcendentals themselves stay one per component.  Component for component the SAME operations as Num<true>: bit-identical values.
struct Num2 {
    static __device__ __forceinline__ f32x2 exp2(f32x2 x) { f32x2 r; r[0] = __builtin_amdgcn_exp2f(x[0]); r[1] = __builtin_amdgcn_exp2f(x[1]); return r; }
    static __device__ __forceinline__ f32x2 log2(f32x2 x) { f32x2 r; r[0] = __builtin_amdgcn_logf(x[0]); r[1] = __builtin_amdgcn_logf(x[1]); return r; }
    static __device__ __forceinline__ f32x2 rcp(f32x2 x) { f32x2 r; r[0] = __builtin_amdgcn_rcpf(x[0]); r[1] = __builtin_amdgcn_rcpf(x[1]); return r; }
    static __device__ __forceinline__ f32x2 abs(f32x2 x) { f32x2 r; r[0] = fabsf(x[0]); r[1] = fabsf(x[1]); return r; }
    static __device__ __forceinline__ f32x2 exp(f32x2 x) { return exp2(x * 1.4426950408889634f); }
    static __device__ __forceinline__ f32x2 ln(f32x2 x) { return log2(x) * 0.6931471805599453f; }
    static __device__ __forceinline__ f32x2 tanh(f32x2 x) {
        const f32x2 r = rcp(1.f + exp2(x * 2.8853900817779268f));
        f32x2 o; o[0] = __builtin_fmaf(-2.f, r[0], 1.f); o[1] = __builtin_fmaf(-2.f, r[1], 1.f);
        return o;
    }
    static __device__ __forceinline__ f32x2 sigmoid(f32x2 x) { return rcp(1.f + exp2(x * -1.4426950408889634f)); }
    static __device__ __forceinline__ f32x2 softplus(f32x2 x) {
        const f32x2 s = log2(1.f + exp2(x * 1.4426950408889634f)) * 0.6931471805599453f;
        f32x2 o; o[0] = x[0] > 20.f ? x[0] : s[0]; o[1] = x[1] > 20.f ? x[1] : s[1];
        return o;
    }
};

// Transcendentals of the BACKWARD recompute.  The forward evaluates the flows with correctly rounded libm calls (parity
// of the rendered values); the backward recomputes the same quantities on the hardware exp / reciprocal (~1 ulp each): tanh =
// 1 - 2 / (1 + e^2x) (saturates correctly at both ends), sigmoid = 1 / (1 + e^-x).  ~8 instead of ~30 instructions per tanh,
// 16 tanh per (sample, k).  (The exp form returns tanh with an ABSOLUTE error of ~1e-7, i.e. a relative one of 1e-7 / |x| near 0;
// round 3 tried the odd Taylor polynomial below |x| = 0.25 - +3 % on the kernel, no gradient of the test suite moved: the
// deviations of the alpha-path gradients it was suspected of are the conditioning of those cases, see tests/util_hip.py.)
__device__ __forceinline__ float t_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float t_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
// (one multiply for the exponent - (2 x) * log2(e) and x * (2 log2(e)) are the same fp32 number, doubling is exact - and one fma for
//  1 - 2 r, likewise the same number as the separate multiply and subtract: 3 + 2 instead of 5 + 2 instructions, bit-identical values)
__device__ __forceinline__ float t_tanh(float x) {
    return __builtin_fmaf(-2.f, t_rcp(1.f + __builtin_amdgcn_exp2f(x * 2.8853900817779268f)), 1.f);
}
__device__ __forceinline__ float t_sigmoid(float x) { return t_rcp(1.f + t_exp(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
// inclusive multiplicative scan across the 64 lanes of a wave
__device__ __forceinline__ float wave_scan_mul(float v) {
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float t = __shfl_up(v, d, 64);
        if (lane >= d) v *= t;
    }
    return v;
}
// ---- the same scans / reductions on DPP (data-parallel primitives: the cross-lane move is a modifier of a vector-ALU instruction,
// no LDS round trip).  __shfl_* compile to ds_bpermute_b32 - an LDS instruction with ~100 cycles of latency - and a scan is a chain
// of six of them; the gfx9 DPP scan is row_shr:1, 2, 4, 8 inside each row of 16 lanes, then row_bcast:15 / row_bcast:31 to carry the
// row totals on (the sequence LLVM's own wave scan uses).  A lane without a source keeps `identity` (bound_ctrl off).
constexpr bool kUseDpp = true;
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp_mov(float identity, float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(src), CTRL, ROW_MASK, BANK_MASK, false));
}
__device__ __forceinline__ float wave_scan_add_dpp(float v) {          // inclusive prefix sum over the 64 lanes
    v += dpp_mov<0x111, 0xf, 0xf>(0.f, v);
    v += dpp_mov<0x112, 0xf, 0xf>(0.f, v);
    v += dpp_mov<0x114, 0xf, 0xf>(0.f, v);
    v += dpp_mov<0x118, 0xf, 0xf>(0.f, v);
    v += dpp_mov<0x142, 0xa, 0xf>(0.f, v);                              // row_bcast:15 -> rows 1, 3
    v += dpp_mov<0x143, 0xc, 0xf>(0.f, v);                              // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ float wave_scan_mul_dpp(float v) {          // inclusive prefix product
    v *= dpp_mov<0x111, 0xf, 0xf>(1.f, v);
    v *= dpp_mov<0x112, 0xf, 0xf>(1.f, v);
    v *= dpp_mov<0x114, 0xf, 0xf>(1.f, v);
    v *= dpp_mov<0x118, 0xf, 0xf>(1.f, v);
    v *= dpp_mov<0x142, 0xa, 0xf>(1.f, v);
    v *= dpp_mov<0x143, 0xc, 0xf>(1.f, v);
    return v;
}
// value of the previous lane (lane 0: `identity`): wave_shr:1
__device__ __forceinline__ float wave_prev_dpp(float v, float identity) { return dpp_mov<0x138, 0xf, 0xf>(identity, v); }
__device__ __forceinline__ float wave_last(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }
__device__ __forceinline__ float wave_sum_dpp(float v) { return wave_last(wave_scan_add_dpp(v)); }     // (same in every lane: an SGPR)

// The cross-lane steps of the composite and of its adjoint, in ONE place: the fused forward / tail kernels and the standalone
// composite_kernel / composite_bwd_kernel must use the same arithmetic (the unfused seam's gradients are held to the fused path's to 1e-6).
__device__ __forceinline__ void comp_scan_mul(float xk, float& incl, float& excl) {     // inclusive / exclusive running product over the lanes
    if (kUseDpp) { incl = wave_scan_mul_dpp(xk); excl = wave_prev_dpp(incl, 1.f); }
    else { incl = wave_scan_mul(xk); excl = __shfl_up(incl, 1, 64); if (lane_id() == 0) excl = 1.f; }
}
__device__ __forceinline__ float comp_sum(float x) { return kUseDpp ? wave_sum_dpp(x) : wave_sum(x); }
__device__ __forceinline__ float comp_last(float incl) { return kUseDpp ? wave_last(incl) : __shfl(incl, 63, 64); }
// ---- the adjoint of the transmittance product (RUN:443), carried as the CANCELLED quantity itself (round 6) --------------------------------
// d loss / d alpha_s = g_s T_s - (sum_{j>s} g_j w_j) / x_s   (g = d loss / d w, x = 1 - alpha + 1e-10, T_{s+1} = T_s x_s: what torch's
// cumprod backward evaluates and what rounds 1-5 evaluated here) is the difference of two terms of size |g| T_s that agree to 2-3 digits
// wherever the samples behind s have the colour of s (any opaque surface).  Every T_j carries its own rounding history: relative to the
// exact product, rho_j.  The computed gradient is the EXACT gradient of sum_j g_j w_j (1 + rho_j) - and how far that is from the exact
// one depends on how rho varies ALONG THE RAY, not on its size: torch's sequential cumprod makes rho a random walk with 3e-8 steps
// (neighbours share all but one rounding: the error is ~rho x the exact gradient), the wave scan of the forward gives every lane its own
// product tree (white noise of ~1e-7: the error is 1e-7 x the UNCANCELLED terms).  Measured on single rays (tests/tools/density_bisect.py,
// profiles/r06_density_bisect.txt): the kernel's formula in fp64 fed the stashed fp32 T reproduces the kernel's error (bias sums of the
// density heads 1e-4 of their largest entry where torch's fp32 autograd is at 3e-6 .. 9e-6) - the product scan's T, nothing else.
// So the recurrence is written for D_s = g_s - R_{s+1}, R_{s+1} = what the samples behind s render for the cotangent g (the quantity
// that is left after the cancellation):
//     d loss / d alpha_s = T_s D_s,      D_s = (g_s - g_{s+1}) + x_{s+1} D_{s+1} - 1e-10 g_{s+1},      D behind the last sample = 0
// (R_{s+1} = g_{s+1} alpha_{s+1} + x_{s+1} R_{s+2} with alpha = 1 - x + 1e-10).  Its inputs are colour DIFFERENCES of neighbouring samples,
// T enters once as a plain factor (its white noise stays relative to the result), nothing is divided by x ~ 1e-10.  In exact arithmetic
// the same number as the suffix form.  One suffix scan of affine maps y -> a y + b over the wave (lane order reversed by one LDS permute,
// then the gfx9 DPP prefix-scan sequence) + (g, x, D) of the first sample of the chunk behind as the carry.
__device__ __forceinline__ void wave_scan_affine_dpp(float& a, float& b) {      // lane r: the composition of the maps of lanes r, r - 1, ... 0
#define CFN_AFF(CTRL, RM) { const float as = dpp_mov<CTRL, RM, 0xf>(1.f, a), bs = dpp_mov<CTRL, RM, 0xf>(0.f, b); b = __builtin_fmaf(a, bs, b); a = a * as; }
    CFN_AFF(0x111, 0xf) CFN_AFF(0x112, 0xf) CFN_AFF(0x114, 0xf) CFN_AFF(0x118, 0xf) CFN_AFF(0x142, 0xa) CFN_AFF(0x143, 0xc)
#undef CFN_AFF
}
// lane = sample of a 64-sample chunk, chunks walked back to front.  g: d loss / d w_s (0 for a lane past the ray's last sample), xk: the
// forward's cumprod factor; car_*: (g, x, D) of the first sample of the chunk behind (0, 0, 0 behind the ray's end), updated to this chunk's.
__device__ __forceinline__ float comp_adjoint_D(const float g, const float xk, float& car_g, float& car_x, float& car_D) {
    static_assert(kUseDpp, "the affine scan is written on DPP");
    const int ridx = (63 - lane_id()) * 4;
    const float rg = __int_as_float(__builtin_amdgcn_ds_bpermute(ridx, __float_as_int(g)));      // reversed: lane r holds sample 63 - r
    const float rx = __int_as_float(__builtin_amdgcn_ds_bpermute(ridx, __float_as_int(xk)));
    const float gn = wave_prev_dpp(rg, car_g), xn = wave_prev_dpp(rx, car_x);                      // sample s + 1
    float a = xn, b = __builtin_fmaf(-1e-10f, gn, rg - gn);
    wave_scan_affine_dpp(a, b);
    const float Dr = __builtin_fmaf(a, car_D, b);
    car_D = wave_last(Dr); car_g = wave_last(rg); car_x = wave_last(rx);
    return __int_as_float(__builtin_amdgcn_ds_bpermute(ridx, __float_as_int(Dr)));
}

// gamma(v): channel c of the encoding of a 3-vector (HLP:42-51): c<3 identity, then per
// frequency f: sin(2^f v) x3, cos(2^f v) x3.  2^f * v is exact in fp32.
__device__ __forceinline__ float enc_channel(const float* v, int c) {
    if (c < 3) return v[c];
    const int q = (c - 3) / 3, d = (c - 3) - 3 * q;       // q = 2*f + (0 sin | 1 cos)
    const float a = v[d] * (float)(1 << (q >> 1));
    float sv, cv;
    sincosf(a, &sv, &cv);          // the SAME routine as the fused kernel's encode_tile: fused and unfused encodings are bit-identical
    return (q & 1) ? cv : sv;
}

// ---- the conditional triangular Sylvester flows for one (point, latent sample) ------------------
// th[84]: rgb D[(i*3+j)*4+f] | 36+ d1[i*4+f] | 48+ d2[i*4+f] | 60+ b[i*4+f] | 72+ alpha d1[f] | 76+ d2[f] | 80+ b[f]
// (diagonals already tanh-ed, MOD:341-348).  z in/out: rgb (3) and alpha (1).  FLW:204-268, MOD:401-413.
template <bool LOGDET, bool FAST = false>
__device__ __forceinline__ void flows_fwd(const float (&th)[84], float (&z)[3], float& a, float& ld_rgb, float& ld_a) {
    using M = Num<FAST>;
    ld_rgb = 0.f; ld_a = 0.f;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const bool odd = f & 1;
        const float zp0 = odd ? z[2] : z[0], zp1 = z[1], zp2 = odd ? z[0] : z[2];
        const float d2_0 = th[48 + 0 + f], d2_1 = th[48 + 4 + f], d2_2 = th[48 + 8 + f];
        const float d1_0 = th[36 + 0 + f], d1_1 = th[36 + 4 + f], d1_2 = th[36 + 8 + f];
        // R2[i][j>i] = D[j][i]  (full_d.transpose, MOD:375);  R1[i][j>i] = D[i][j] (MOD:374)
        const float pre0 = ((d2_0 * zp0 + th[(1 * 3 + 0) * 4 + f] * zp1) + th[(2 * 3 + 0) * 4 + f] * zp2) + th[60 + 0 + f];
        const float pre1 = (d2_1 * zp1 + th[(2 * 3 + 1) * 4 + f] * zp2) + th[60 + 4 + f];
        const float pre2 = d2_2 * zp2 + th[60 + 8 + f];
        const float t0 = M::tanh(pre0), t1 = M::tanh(pre1), t2 = M::tanh(pre2);
        const float u0 = (d1_0 * t0 + th[(0 * 3 + 1) * 4 + f] * t1) + th[(0 * 3 + 2) * 4 + f] * t2;
        const float u1 = d1_1 * t1 + th[(1 * 3 + 2) * 4 + f] * t2;
        const float u2 = d1_2 * t2;
        z[0] = (odd ? u2 : u0) + z[0];
        z[1] = u1 + z[1];
        z[2] = (odd ? u0 : u2) + z[2];
        const float ta = M::tanh(th[76 + f] * a + th[80 + f]);
        a = th[72 + f] * ta + a;
        if (LOGDET) {
            ld_rgb += (M::ln(fabsf((1.f - t0 * t0) * (d1_0 * d2_0) + 1.f) + 1e-08f) +
                       M::ln(fabsf((1.f - t1 * t1) * (d1_1 * d2_1) + 1.f) + 1e-08f)) +
                      M::ln(fabsf((1.f - t2 * t2) * (d1_2 * d2_2) + 1.f) + 1e-08f);
            ld_a += M::ln(fabsf((1.f - ta * ta) * (th[72 + f] * th[76 + f]) + 1.f) + 1e-08f);
        }
    }
}

// (p[h], p[h]) for a constant h: folds into the op_sel bits of the packed instruction that reads it
__device__ __forceinline__ f32x2 pair_half(const f32x2 p, const int h) { const float v = h ? p[1] : p[0]; f32x2 r; r[0] = v; r[1] = v; return r; }
// the same flows for TWO latent samples of a point (components of the f32x2 values), on Num2: operation for operation flows_fwd<LOGDET, true>.
// The flow parameters come as the 42 register PAIRS tp[j] = (th[2 j], th[2 j + 1]): a packed instruction takes either half of a pair for both
// of its components (op_sel), so th[j] * (z_a, z_b) needs no broadcast copy - a per-element splat into a pair of its own cost 84 more registers.
template <bool LOGDET>
__device__ __forceinline__ void flows_fwd2(const f32x2 (&tp)[42], f32x2 (&z)[3], f32x2& a, f32x2& ld_rgb, f32x2& ld_a) {
    using M = Num2;
#define CFN_TH(j) pair_half(tp[(j) >> 1], (j) & 1)
    ld_rgb = 0.f; ld_a = 0.f;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const bool odd = f & 1;
        const f32x2 zp0 = odd ? z[2] : z[0], zp1 = z[1], zp2 = odd ? z[0] : z[2];
        const f32x2 d2_0 = CFN_TH(48 + 0 + f), d2_1 = CFN_TH(48 + 4 + f), d2_2 = CFN_TH(48 + 8 + f);
        const f32x2 d1_0 = CFN_TH(36 + 0 + f), d1_1 = CFN_TH(36 + 4 + f), d1_2 = CFN_TH(36 + 8 + f);
        const f32x2 pre0 = ((d2_0 * zp0 + CFN_TH((1 * 3 + 0) * 4 + f) * zp1) + CFN_TH((2 * 3 + 0) * 4 + f) * zp2) + CFN_TH(60 + 0 + f);
        const f32x2 pre1 = (d2_1 * zp1 + CFN_TH((2 * 3 + 1) * 4 + f) * zp2) + CFN_TH(60 + 4 + f);
        const f32x2 pre2 = d2_2 * zp2 + CFN_TH(60 + 8 + f);
        const f32x2 t0 = M::tanh(pre0), t1 = M::tanh(pre1), t2 = M::tanh(pre2);
        const f32x2 u0 = (d1_0 * t0 + CFN_TH((0 * 3 + 1) * 4 + f) * t1) + CFN_TH((0 * 3 + 2) * 4 + f) * t2;
        const f32x2 u1 = d1_1 * t1 + CFN_TH((1 * 3 + 2) * 4 + f) * t2;
        const f32x2 u2 = d1_2 * t2;
        z[0] = (odd ? u2 : u0) + z[0];
        z[1] = u1 + z[1];
        z[2] = (odd ? u0 : u2) + z[2];
        const f32x2 ta = M::tanh(CFN_TH(76 + f) * a + CFN_TH(80 + f));
        a = CFN_TH(72 + f) * ta + a;
        if (LOGDET) {
            // d1 d2 of flows (f, f ^ 1) as ONE packed product of two parameter pairs; its half f & 1 serves both latents
            const f32x2 dd0 = tp[(36 + 0 + f) >> 1] * tp[(48 + 0 + f) >> 1], dd1 = tp[(36 + 4 + f) >> 1] * tp[(48 + 4 + f) >> 1];
            const f32x2 dd2 = tp[(36 + 8 + f) >> 1] * tp[(48 + 8 + f) >> 1], dda = tp[(72 + f) >> 1] * tp[(76 + f) >> 1];
#define CFN_HALF(v) pair_half(v, f & 1)
            ld_rgb += (M::ln(M::abs((1.f - t0 * t0) * CFN_HALF(dd0) + 1.f) + 1e-08f) +
                       M::ln(M::abs((1.f - t1 * t1) * CFN_HALF(dd1) + 1.f) + 1e-08f)) +
                      M::ln(M::abs((1.f - t2 * t2) * CFN_HALF(dd2) + 1.f) + 1e-08f);
            ld_a += M::ln(M::abs((1.f - ta * ta) * CFN_HALF(dda) + 1.f) + 1e-08f);
#undef CFN_HALF
        }
    }
#undef CFN_TH
}

// ---- LDS-DMA helpers (the fp32 big-tile loader, the small-job kernel, the standalone composite kernels)
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x4 ds_rsrc(const float* base, int bytes) {
    const unsigned long long p = (unsigned long long)base;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
    r[1] = __builtin_amdgcn_readfirstlane((int)((p >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000;
    return r;
}
// one 1-KB LDS-DMA piece: lane l's 16 bytes at (descriptor base + soff + voff) land at lds_addr + 16 l.  M0 is written in the
// statement that reads it and restored (it is compiler-reserved).
__device__ __forceinline__ void ds_dma16(i32x4 rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// ---- a wave's private LDS block as a turntable (standalone composite kernels: cfnerf_fwd.hip / cfnerf_bwd.hip)
// A wave's own LDS queue is in order; this only keeps the COMPILER from moving LDS accesses across the point where lanes exchange data.
__device__ __forceinline__ void wave_lds_turn() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// The [64 samples][KG latents] block of quads (16 bytes = one (sample, latent) entry of raw [N,S,K,4]) of a 64-sample chunk, filled by
// KG LDS-DMA pieces: piece j, lane l lands at quad position p = 64 j + l.  Position p = sl * KG + c holds latent kk = c ^ swz(sl) of
// sample sl: the KG lanes of a sample still cover its one 16 KG-byte segment of global memory (every piece is 1 KB of full cache lines
// when K is a multiple of KG), and the transposed read - lane = sample, one latent - touches 16 different bank quads per 16 lanes.
template <int KG> struct CompStage {
    static_assert(KG == 4 || KG == 8, "a sample's segment is 64 or 128 bytes");
    static constexpr int kQuads = 64 * KG;
    static __device__ __forceinline__ int swz(int sl) { return (sl / (16 / KG)) & (KG - 1); }
    // this lane's byte offset (inside the chunk's rows of the ray, group 0) of the quad it moves in piece j
    static __device__ __forceinline__ unsigned piece_voff(int lane, int j, int K) {
        const int p = j * 64 + lane, sl = p / KG, kk = (p % KG) ^ swz(sl);
        return (unsigned)((sl * K + kk) * 16);
    }
    static __device__ __forceinline__ void fetch(i32x4 rsrc, unsigned lds0, const unsigned (&voff)[KG], int ch, int g0, int K) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the previous block's LDS reads have their data
        wave_lds_turn();
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((ch * 64 * K + g0) * 16);
#pragma unroll
        for (int j = 0; j < KG; ++j) ds_dma16(rsrc, lds0 + j * 1024, voff[j], soff);
    }
    static __device__ __forceinline__ void landed() {            // (hipcc does not count asm memory operations: explicit wait)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_lds_turn();
    }
};

}  // namespace cfnerf
