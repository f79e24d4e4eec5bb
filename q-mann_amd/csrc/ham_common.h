// ham_common.h -- lane arithmetic of the Hamming-family attention scores, shared by the streaming
// kernel (batch_hops_ham.hip) and the one-wavefront kernel for short memories (hops_small.h).
// See batch_hops_ham.hip for what each mode computes and the reference lines it follows.
#pragma once
#include "hops_common.h"

namespace {

// kModeV0 / kModeV1 read packed bit planes; kModeV0Bytes / kModeV1Bytes compute the same scores straight from
// the sign-magnitude int8 memories (no packing pass; the choice for short memories and for num_bit = 8, where
// planes are no smaller than bytes).  3 = kModeFixed (hops_small.h).
enum { kModeAppx = 0, kModeV0 = 1, kModeV1 = 2, kModeV0Bytes = 4, kModeV1Bytes = 5 };
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

// 12 VALU operations per 4 columns in the compiled loop (the boolean pairs fuse into v_bitop3_b32; round 1: 15 + 2 per row,
// round 2: 13).  Padding columns are forced to "same sign, both magnitudes 0" by the masks (a term of exactly 127), which
// `bias` leaves out.
__device__ __forceinline__ int appx_lane_sum(const i32x4 x, const AppxConst &c)
{
    int dot = 0;
    uint32_t sad = 0;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const uint32_t w = (uint32_t)x[d];
        const uint32_t km = w & c.m7[d];
        const uint32_t sd = (w ^ c.us[d]) & c.m8[d];                        // signs differ
        const uint32_t dmask = __builtin_amdgcn_perm(0u, 0u, sd);           // 0xFF in those bytes
        // same sign: 127 - |ka - kb|; `bias` holds a 127 for every real column.  Where the signs differ the key byte is
        // replaced by |u| + 127: a difference of exactly 127, which gives that column's 127 back inside the same v_sad_u8
        // (round 2 spent a v_dot4 on counting those columns)
        sad = __builtin_amdgcn_sad_u8((km & ~dmask) | (c.u127[d] & dmask), c.um[d], sad);
        // opposite sign: +-(127 - ((ka + kb) & 127)); per byte ka + kb <= 254, no carry across bytes
        const uint32_t s4 = km + c.um[d];                                   // bit 7 of a byte = carry out of 7 bits
        const uint32_t val = ~s4 & (dmask & 0x7F7F7F7Fu);
        const uint32_t ge = km + c.nb[d];                                   // bit 7: |k| + 128 - |u| >= 128, i.e. |k| >= |u| (<= 255: no carry)
        const uint32_t lneg = (w & ge) | (c.us[d] & ~ge);                   // bit 7: sign of the larger operand
        const uint32_t neg = ((s4 & lneg) | ~s4) & 0x80808080u;             // negative unless carry and larger > 0
        const uint32_t sg = __builtin_amdgcn_perm(0x01010101u, 0x01010101u, neg);
        dot = __builtin_amdgcn_sdot4((int)val, (int)sg, dot, false);
    }
    return dot + c.bias - (int)sad;
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
