// ham_common.h -- lane arithmetic of the Hamming-family attention scores, shared by the streaming
// kernel (batch_hops_ham.hip) and the one-wavefront kernel for short memories (hops_small.h).
// See batch_hops_ham.hip for what each mode computes and the reference lines it follows.
#pragma once
#include "hops_common.h"

namespace {

// kModeV0 / kModeV1 read packed bit planes; kModeV0Bytes / kModeV1Bytes compute the same scores straight from
// the sign-magnitude int8 memories (no packing pass; the choice for short memories and for num_bit = 8, where
// planes are no smaller than bytes).  3 = kModeFixed (hops_small.h).
// kModeAppxMq: APPX whose hops differ in operand grids (mixed quantisation, see "APPX under EN_MQ" below); kModeAppx is the
// same kernel with every hop of kind kHamSame compiled in.
enum { kModeAppx = 0, kModeV0 = 1, kModeV1 = 2, kModeV0Bytes = 4, kModeV1Bytes = 5, kModeAppxMq = 6 };
constexpr bool mode_is_appx(int m) { return m == kModeAppx || m == kModeAppxMq; }
constexpr bool mode_is_v0(int m) { return m == kModeV0 || m == kModeV0Bytes; }
constexpr bool mode_is_planes(int m) { return m == kModeV0 || m == kModeV1; }

// V0 scores are small counts (0 .. num_bit . D <= 2048): the softmax is evaluated once per distinct
// count through a histogram (as the fixed-point kernel does per code).  The tables (hist u32, p float,
// Q(p) u8, each nbins = num_bit . D + 1 entries) sit behind the score array.
__host__ __device__ inline uint32_t v0_hist_bytes(uint32_t nbins) { return ((nbins * 4 + 15) / 16) * 16; }
__host__ __device__ inline uint32_t v0_table_bytes(uint32_t nbins) { return 2 * v0_hist_bytes(nbins) + ((nbins + 15) / 16) * 16; }

// ---- APPX: 4 key bytes against 4 query bytes -------------------------------------------------
struct AppxConst {
    uint32_t um[4];   // |u| bytes
    uint32_t u127[4]; // |u| + 127 per byte (<= 254): what a sign-differing byte is replaced by in the same-sign sum
    uint32_t nb[4];   // 128 - |u| per byte
    uint32_t us[4];   // 0x80 where u < 0
    uint32_t m7[4];   // 0x7F in real columns, 0 in padding
    uint32_t m8[4];   // 0x80 in real columns, 0 in padding
    int bias;         // 127 . (16 - padding columns of this lane): the same-sign terms' constant part
};

// APPX under EN_MQ (mixed quantisation, MemN2N/MemN2N.c:748-754: weight formats Q(iwl+1.frac-1), Q(iwl.frac), Q(iwl-1.frac+1)
// for hops 0, 1, 2 while the attention format stays Q(iwl.frac)).  The reference re-encodes both operands of a column as
// Q(iwl_att, 31 - iwl_att) sign-magnitude WORDS (lib/layer_cuda.cu:355-420) and only bits 30..24 of the adjusted words are
// compared, but the adjustment (common magnitude removed / moved onto the larger operand) runs on all 31 bits.  What a byte per
// operand can carry exactly, by the hop's KIND (qfmt.h::ham_hop_kind):
//   kHamSame    both operand grids lie inside the attention grid: the word is byte << 24 (the case of the rounds before).
//   kHamCoarse  both grids are at least one fractional bit coarser, wider range: on-grid codes are even, so the odd code 127 is
//               free and stands for the SATURATED word 0x7FFFFFFF (Q_att clamps to 127 by itself).  Its low 24 ones never
//               matter -- same sign: 0x7FFFFFFF - (b << 24) has top bits 127 - b; opposite: + (b << 24) carries nothing up --
//               except when BOTH operands are saturated with opposite signs: 0x7FFFFFFF + 0x7FFFFFFF = 0xFFFFFFFE, top bits
//               carry + 127 where 127 + 127 gives carry + 126.  `km & um & 1` is that case (every other code is even).
//   kHamFine    the key grid is one fractional bit finer and at least one integer bit narrower; u lies inside the attention
//               grid.  The key byte is the key's OWN code k8 (units 2^-(frac+1), <= 127), u's word is (2 u7) << 23:
//                 opposite sign  top bits of k8 + 2 u7 are (k8 >> 1) + u7, carry and comparison likewise: the byte
//                                arithmetic on kh = k8 >> 1;
//                 same sign      |k8 - 2 u7| >> 1 = |kh - u7| - [k8 odd and kh < u7].
// A value of exactly -2^iwl_att (reachable on a wider grid only) is "minus zero" in the reference's word: byte 0x80
// (ham_ubyte below; the embedding kernels' key bytes follow the same rule).
__device__ __forceinline__ uint32_t ham_kind_of(const HopArgs &a, uint32_t h) { return (a.ham_kinds >> (2u * h)) & 3u; }

// operations per 4 columns in the compiled loop: kHamSame 12 (the boolean pairs fuse into v_bitop3_b32; round 1: 15 + 2 per
// row, round 2: 13), kHamCoarse 13, kHamFine 17.  Padding columns are forced to "same sign, both magnitudes 0" by the masks
// (a term of exactly 127), which `bias` leaves out.
template <int KIND>
__device__ __forceinline__ int appx_lane_sum_k(const i32x4 x, const AppxConst &c)
{
    int dot = 0;
    uint32_t sad = 0, fix = 0;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const uint32_t w = (uint32_t)x[d];
        const uint32_t km = KIND == kHamFine ? ((w >> 1) & ((c.m7[d] >> 1) & 0x3F3F3F3Fu)) : (w & c.m7[d]);
        const uint32_t sd = (w ^ c.us[d]) & c.m8[d];                        // signs differ
        const uint32_t dmask = __builtin_amdgcn_perm(0u, 0u, sd);           // 0xFF in those bytes
        // same sign: 127 - |ka - kb|; `bias` holds a 127 for every real column.  Where the signs differ the key byte is
        // replaced by |u| + 127: a difference of exactly 127, which gives that column's 127 back inside the same v_sad_u8
        // (round 2 spent a v_dot4 on counting those columns)
        sad = __builtin_amdgcn_sad_u8((km & ~dmask) | (c.u127[d] & dmask), c.um[d], sad);
        // opposite sign: +-(127 - ((ka + kb) & 127)); per byte ka + kb <= 254 (+ 1 for two saturated operands), no carry across bytes
        const uint32_t s4 = KIND == kHamCoarse ? km + c.um[d] + (km & c.um[d] & 0x01010101u) : km + c.um[d];   // bit 7 of a byte = carry out of 7 bits
        const uint32_t val = ~s4 & (dmask & 0x7F7F7F7Fu);
        const uint32_t ge = km + c.nb[d];                                   // bit 7: |k| + 128 - |u| >= 128, i.e. |k| >= |u| (<= 255: no carry)
        const uint32_t lneg = (w & ge) | (c.us[d] & ~ge);                   // bit 7: sign of the larger operand
        const uint32_t neg = ((s4 & lneg) | ~s4) & 0x80808080u;             // negative unless carry and larger > 0
        const uint32_t sg = __builtin_amdgcn_perm(0x01010101u, 0x01010101u, neg);
        dot = __builtin_amdgcn_sdot4((int)val, (int)sg, dot, false);
        if (KIND == kHamFine) {
            // same sign, odd key code, kh < u7: the halved difference is one less than |kh - u7| (128 per such column)
            const uint32_t odd7 = (w << 7) & c.m8[d];                       // bit 7 of a byte = bit 0 of that key byte
            fix = __builtin_amdgcn_sad_u8(odd7 & ~ge & ~sd, 0u, fix);
        }
    }
    return dot + c.bias - (int)sad + (int)(fix >> 7);
}
__device__ __forceinline__ int appx_lane_sum(const i32x4 x, const AppxConst &c) { return appx_lane_sum_k<kHamSame>(x, c); }

// sign-magnitude byte of the attention operand u of a Hamming-family mode: magnitude Q_att (clamped at 127, truncated toward
// zero), sign from the VALUE; exactly -2^iwl is the reference's "minus zero" (its operand word is Q(iwl, 31 - iwl):
// lib/layer_cuda.h:233-253 -- the value is not below the macro's float limit, converts to INT32_MIN, whose sign-magnitude
// word has magnitude 0).  Only reachable when u's own grid is wider than the attention grid.
__device__ __forceinline__ uint32_t ham_ubyte(float ua, QFmt fm, bool real)
{
    if (!real) return 0u;
    const int kc = qm_code(ua, fm.iwl, fm.frac);
    const uint32_t mag = ua == -(float)(1u << fm.iwl) ? 0u : (uint32_t)(kc < 0 ? -kc : kc);
    return mag | (!(ua >= 0.0f) ? 0x80u : 0u);
}

// Final quantisation of a mode-3 score, Q(iwl, 31-iwl) (lib/layer_cuda.cu:2515): saturation at +-2^iwl -- except that a sum of
// EXACTLY -2^iwl is not below the macro's float limit (the limit (2^31 - 1) / 2^(31-iwl) rounds to 2^iwl itself), converts to
// INT32_MIN, and the sign-magnitude word of INT32_MIN is "minus zero": the reference returns 0 there
// (lib/layer_cuda.h:233-253; +2^iwl converts to INT32_MAX and stays 2^iwl).  v and lim in units of 2^-10.
__device__ __forceinline__ int appx_clamp(int v, int lim)
{
    return v > lim ? lim : (v < -lim ? -lim : (v == -lim ? 0 : v));
}

// ---- V0 / V1: two 64-bit plane words per lane -------------------------------------------------
struct PlaneConst {
    uint64_t u[2];      // query plane words at this lane's (group, plane) positions
    uint64_t valid[2];  // real-column mask of the word's group
    uint64_t us[2];     // query SIGN plane of the word's group
    int wgt[2];         // V1 weight 2^(n-1-i) of the word's plane (0 for the sign plane)
};

template <int MODE, int NB>
__device__ __forceinline__ int plane_lane_sum(const i32x4 x, const PlaneConst &c)
{
    uint64_t k[2];
    k[0] = (uint64_t)(uint32_t)x[0] | ((uint64_t)(uint32_t)x[1] << 32);
    k[1] = (uint64_t)(uint32_t)x[2] | ((uint64_t)(uint32_t)x[3] << 32);
    if (MODE == kModeV0) {
        return __popcll(~(k[0] ^ c.u[0]) & c.valid[0]) + __popcll(~(k[1] ^ c.u[1]) & c.valid[1]);
    }
    // V1: the key's sign plane of each word's group sits in the lane that holds plane 0 of that
    // group: this lane (NB <= 2), the even lane of the pair (NB == 4) or the first lane of the
    // quad (NB == 8) -- fetched with quad-permute DPP moves
    uint64_t ks[2];
    if (NB == 1) {
        return 0;                                   // no magnitude planes: every weight is zero
    } else if (NB == 2) {
        ks[0] = k[0]; ks[1] = k[0];
    } else {
        constexpr int ctrl = (NB == 4) ? 0xA0 /* quad_perm [0,0,2,2] */ : 0x00 /* [0,0,0,0] */;
        const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, x[0], ctrl, 0xF, 0xF, true);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, x[1], ctrl, 0xF, 0xF, true);
        ks[0] = (uint64_t)lo | ((uint64_t)hi << 32);
        ks[1] = ks[0];
    }
    int acc = 0;
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const uint64_t eq = ~(k[t] ^ c.u[t]);
        const uint64_t sdiff = (ks[t] ^ c.us[t]) & c.valid[t];
        const uint64_t ssame = ~(ks[t] ^ c.us[t]) & c.valid[t];
        acc += c.wgt[t] * (__popcll(eq & ssame) - __popcll(eq & sdiff));
    }
    return acc;
}


// ---- V0 / V1 on int8 keys ---------------------------------------------------------------------------
// The top num_bit bits of a sign-magnitude byte ARE planes 0..num_bit-1 of that column, so
//   V0 = sum over real columns of (n - popcount((k ^ u) & top_n))                      (lib/common.c:223-246)
//   V1 = sum over real columns of sgn.sgn.( (2^(n-1) - 1) - (((|k| ^ |u|) >> (8 - n)) & (2^(n-1) - 1)) )
//        in units of 2^-n: the weights 2^(n-1-i) of bits i = 1..n-1 add up to that difference (lib/common.c:249-312)
struct HamByteConst {
    uint32_t ub[4];   // sign-magnitude bytes of Q_att(u) for this lane's 16 columns
    uint32_t m[4];    // V0: top-n-bit mask, V1: low (n-1)-bit mask; 0 in padding columns
    int bias;         // V0: n . (real columns of this lane)
};
template <int MODE, int NB>
__device__ __forceinline__ void make_hambyte_const(HamByteConst &c, const uint8_t *ub, uint32_t c0, uint32_t D)
{
    constexpr uint32_t pat = mode_is_v0(MODE) ? ((0xFF00u >> NB) & 0xFFu) : ((1u << (NB - 1)) - 1u);
#pragma unroll
    for (int d = 0; d < 4; d++) {
        c.ub[d] = *(const uint32_t *)(ub + c0 + 4 * d);
        uint32_t vm = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) vm |= (c0 + 4 * d + i < D ? pat : 0u) << (8 * i);
        c.m[d] = vm;
    }
    c.bias = NB * (int)(D >= c0 + 16 ? 16u : (D > c0 ? D - c0 : 0u));
}
template <int MODE, int NB>
__device__ __forceinline__ int hambyte_lane_sum(const i32x4 x, const HamByteConst &c)
{
    if (mode_is_v0(MODE)) {
        uint32_t diff = 0;
#pragma unroll
        for (int d = 0; d < 4; d++) diff += (uint32_t)__builtin_popcount(((uint32_t)x[d] ^ c.ub[d]) & c.m[d]);
        return c.bias - (int)diff;
    }
    if (NB == 1) return 0;                          // no magnitude bits compared: every weight is zero
    int dot = 0;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const uint32_t xo = (uint32_t)x[d] ^ c.ub[d];
        const uint32_t y = ((xo & 0x7F7F7F7Fu) >> (8 - NB));                // per byte: the compared magnitude bits that differ
        const uint32_t t = ~y & c.m[d];                                     // W - y, 0 in padding columns
        const uint32_t sg = __builtin_amdgcn_perm(0x01010101u, 0x01010101u, xo & 0x80808080u);   // -1 where the signs differ
        dot = __builtin_amdgcn_sdot4((int)t, (int)sg, dot, false);
    }
    return dot;
}

// AppxConst of the 16 columns starting at c0, from the sign-magnitude Q_att(u) bytes `ub`
__device__ __forceinline__ void make_appx_const(AppxConst &c, const uint8_t *ub, uint32_t c0, uint32_t D)
{
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const uint32_t b4 = *(const uint32_t *)(ub + c0 + 4 * d);
        c.um[d] = b4 & 0x7F7F7F7Fu;
        c.u127[d] = (b4 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;                         // per byte <= 254: no carry
        c.nb[d] = (~b4 & 0x7F7F7F7Fu) + 0x01010101u;                          // (127 - |u|) + 1 per byte
        c.us[d] = b4 & 0x80808080u;
        uint32_t vm = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) vm |= (c0 + 4 * d + i < D ? 0xFFu : 0u) << (8 * i);
        c.m7[d] = vm & 0x7F7F7F7Fu;
        c.m8[d] = vm & 0x80808080u;
    }
    c.bias = 127 * (int)(D >= c0 + 16 ? 16u : (D > c0 ? D - c0 : 0u));
}

// PlaneConst of the two plane words a lane owns (word index 2.chunk + t inside the row), from the
// query planes `upl` ([group][8] uint64)
template <int NB>
__device__ __forceinline__ void make_plane_const(PlaneConst &c, const uint64_t *upl, uint32_t chunk, uint32_t D)
{
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const uint32_t wi = 2 * chunk + t;
        const uint32_t grp = wi / NB, pl = wi % NB;
        c.u[t] = upl[grp * 8 + pl];
        c.us[t] = upl[grp * 8 + 0];
        const uint32_t ncol = D > 64 * grp ? D - 64 * grp : 0;
        c.valid[t] = ncol >= 64 ? ~0ull : ((1ull << ncol) - 1ull);
        c.wgt[t] = pl == 0 ? 0 : (1 << (NB - 1 - pl));
    }
}

}  // namespace
