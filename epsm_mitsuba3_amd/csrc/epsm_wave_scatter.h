// epsm_wave_scatter.h -- wave64 merge of scatter items before the float atomics.
//
// Global float atomics on MI355X execute at the memory side at one chip-wide
// rate and serialise per address (MI355X_MICROARCH.md, "Global float atomics"):
// 64 lanes adding to 64 unrelated rows, or many waves adding to the same row,
// are an order of magnitude below the contiguous rate.  Neighbouring paths of a
// wavefront hit the same triangle most of the time (they are samples of the
// same / adjacent pixels), so before any atomic the items of a wave are merged:
//   * stand-alone scatter kernel: runs of adjacent lanes with an identical key
//     triple are summed with a segmented shuffle scan when the wave has few
//     runs; only the last lane of a run issues the atomic;
//   * fused kernel: lanes with the same target (one pixel's samples on the first
//     triangle, the couple of emitter triangles, a BSDF's alpha slot), adjacent
//     or not, are summed with DPP adds in bounded leader rounds
//     (merge_equal below; epsm_grad_scatter.hip, epsm_backward_cp.hip).
// Device-only code.
#pragma once

#include <hip/hip_runtime.h>
#include "epsm_path_core.h"

namespace epsm {

struct Runs {
    int head;        // lane index of the first lane of this lane's run
    bool tail;       // this lane is the last of its run and the run is valid
};

__device__ __forceinline__ int lane_id() { return __lane_id(); }

// Runs of adjacent lanes that are `valid` and share the same three keys.
__device__ __forceinline__ Runs make_runs(bool valid, uint32_t k0, uint32_t k1, uint32_t k2) {
    const int lane = lane_id();
    const uint32_t p0 = __shfl_up(k0, 1), p1 = __shfl_up(k1, 1), p2 = __shfl_up(k2, 1);
    const int pv = __shfl_up((int) valid, 1);
    const bool head = (lane == 0) || !pv || !valid || p0 != k0 || p1 != k1 || p2 != k2;
    const unsigned long long heads = __ballot(head);
    const unsigned long long below = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
    Runs r;
    r.head = 63 - __clzll((long long) (heads & below));
    r.tail = valid && ((lane == 63) || ((heads >> (lane + 1)) & 1ull));
    return r;
}

// Inclusive segmented sum: after the call the tail lane of every run holds the run total.
__device__ __forceinline__ float seg_sum(float v, int head) {
    const int lane = lane_id();
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float t = __shfl_up(v, off);
        if (lane - off >= head) v += t;
    }
    return v;
}
__device__ __forceinline__ V3<float> seg_sum3(V3<float> v, int head) {
    return mk3<float>(seg_sum(v.x, head), seg_sum(v.y, head), seg_sum(v.z, head));
}

// ---------------------------------------------------------------------------
// Workgroup-private accumulator in LDS (open addressing, linear probing).
//
// Same-address global float atomics serialise at ~25 ns each, so keys that the whole
// wavefront hits (the couple of emitter triangles every light sample lands on, the
// per-BSDF alpha slots, a big triangle seen by thousands of pixels) must not cost one
// global atomic per wave.  A persistent workgroup sums into this table with LDS
// atomics and flushes each live row to HBM once per fill (and once at the end).
// Key space: row r of buffer `which` (0 pos, 1 nrm) -> which*V + r; alpha slot b -> 2V + b.
// ---------------------------------------------------------------------------
constexpr uint32_t kEmptyKey = 0xFFFFFFFFu;
// Buckets an insertion tries before its row leaves for the buffers directly.  8 while such a row cost three atomic requests; since
// drain_queue sends it as one (four lanes per row), a crowded neighbourhood is better left early -- headline / pool / dense
// specular slab, ms: 8 probes 1.82 / 2.38 / 6.56, 16: 1.90 / 2.61 / 7.72, 4: 1.79 / 2.27 / 6.17, 3: 1.79 / 2.18* / 7.09, 2: 1.79 /
// 2.25 / 24.8 (the table no longer holds the rows the window shares)   (*: with windows of 2048 paths)
#ifndef EPSM_MAX_PROBE
#define EPSM_MAX_PROBE 4
#endif
constexpr int kMaxProbe = EPSM_MAX_PROBE;

// What a table row sums in.  float: 12 B of values per row, ds_add_f32.  Fixed64: 64-bit fixed point with 44 fractional
// bits, 24 B per row, ds_add_u64: a resolution of 5.7e-14 -- float32's own on terms down to 1e-6, the terms being clamped to
// +-clip (0.1, epsm.py:932-944) and ending up in float32 buffers -- a range of +-2^19 for the sum of a row between two
// flushes (a table lives for a few thousand paths), and a sum that does not depend on the order of the additions.
// The LDS integer atomic is ~15x faster than the float one (rates below) and does not serialise as badly on one address,
// but a row is 1.75x as large.  The fused kernels use fixed-point rows whenever the clamp is on (clip <= 1); a caller who
// switched it off gets float rows (unbounded terms).  (Round 2 had 32 fractional bits: 2.3e-10, i.e. 2e-4 of a term of
// 1e-6 -- visible in a per-path parity test with small tangents, tests/test_gpu_backward_per_path.py.)
struct AccFloat {
    typedef float T;
    static constexpr bool kBucketed = false;         // see LdsTable::add
    __device__ __forceinline__ static bool fits(float, float, float) { return true; }
    __device__ __forceinline__ static void add(T *p, float x) { if (fabsf(x) < __builtin_inff()) atomicAdd(p, x); }   // non-finite terms add nothing (AccFixed64::to_fixed)
    __device__ __forceinline__ static float get(T q) { return q; }
};
__device__ __forceinline__ bool adds_something(float x) { return x != 0.f && fabsf(x) < __builtin_inff(); }   // neither 0 nor NaN / inf
struct AccFixed64 {
    typedef long long T;
    static constexpr bool kBucketed = true;
    // Terms below this magnitude go into the LDS rows; anything else -- a product with an unbounded emitter weight, inf, NaN --
    // takes LdsTable::global_add (exact float atomics; non-finite terms add nothing: the reference's buffers would hold NaN
    // there and its optimiser loop scrubs that to 0, EPSM/optim.py:143-154).  Terms are clamped to +-clip <= 1 before any
    // weight multiplies them, so the slow way is taken by emitter weights beyond ~10^2 only.
    // The limit also bounds the SUM: a row holds |sum| < 2^19; between two flushes a table sees at most one window
    // of <= 2048 paths (kernel: the table is flushed after every window when a workgroup walks several), and a path adds at
    // most 3 K + 1 = 16 terms to one row (position, end-point and emitter rows of its K <= 5 vertices + the occluder), each
    // < 16 in magnitude: 2048 x 16 x 16 = 2^19, never reached -- the int64 cannot wrap.
    static constexpr float kLimit = 16.f;
    __device__ __forceinline__ static bool fits(float x, float y, float z) {
        return fabsf(x) < kLimit && fabsf(y) < kLimit && fabsf(z) < kLimit;        // three compares, each false on NaN (a max3 would skip it)
    }
    __device__ __forceinline__ static T to_fixed(float x) {
        // round(x * 2^44) for |x| < 2^7 (kLimit keeps it below 2^4) in THREE instructions: x 2^44 + 1.5 2^52 in float64 (one fma, round to nearest) keeps
        // the exponent of the constant, so the integer sits in the mantissa in two's complement and the constant's bit
        // pattern -- whose low word is zero -- comes off with one 32-bit subtraction.  (Round 3 split |x| 2^12 into integer
        // part and fraction and negated in 64 bits: ~15 instructions per value, 45 per row, the larger half of a drain
        // iteration; there is no float -> int64 instruction.  Headline slab 2.10 -> 2.07 ms, config 2 2.53 -> 2.46.)
        const double y = __builtin_fma((double) x, 17592186044416.0, 6755399441055744.0);
        return __double_as_longlong(y) - 0x4338000000000000LL;
    }
    __device__ __forceinline__ static void add(T *p, float x) { atomicAdd((unsigned long long *) p, (unsigned long long) to_fixed(x)); }
    __device__ __forceinline__ static float get(T q) { return (float) ((double) q * 5.6843418860808015e-14); }      // 2^-44
};

// (float64 rows -- ds_add_f64, one v_cvt_f64_f32 per term instead of the ten instructions of to_fixed, the LDS atomic at half
// the rate of ds_add_u64: 3.1 against 6.1 lane-operations per clock and CU, tools/micro/lds_atomics.hip -- measured the same
// kernel time, 2.168 against 2.171 ms on the headline slab: the drain is bound by neither.  Not kept: fixed-point sums do not
// depend on the order of the additions.)
template <int kRows, typename Acc = AccFloat>
struct LdsTable {
    static constexpr int kTableSize = kRows;             // rows: 4 B key + 3 values each; any count (multiply-shift hash)
    typedef typename Acc::T Val;
    uint32_t *keys;      // [kTableSize]
    Val *vals;           // [kTableSize][3]
    int *used;           // number of occupied rows
    float *gpos, *gnrm, *galpha;
    uint32_t V;

    __device__ __forceinline__ void global_add(uint32_t key, float x, float y, float z) const {
        float *p;
        if (key < V) p = gpos + 3 * (int64_t) key;
        else if (key < 2u * V) p = gnrm + 3 * (int64_t) (key - V);
        else { if (adds_something(x)) atomicAdd(galpha + (key - 2u * V), x); return; }
        if (adds_something(x)) atomicAdd(p + 0, x);
        if (adds_something(y)) atomicAdd(p + 1, y);
        if (adds_something(z)) atomicAdd(p + 2, z);
    }
    // LDS atomic rates on MI355X (tools/micro/lds_atomics.hip; lane-ops per clock and CU, scattered rows / 16 hot rows):
    // ds_add_f32 0.33 / 0.42, read + ds_cmpst float-add loop 3.3 / 0.47, ds_add_u64 6.1 / 3.4, ds_add_u32 11.4 / 4.3,
    // ds_cmpst_rtn 5.6.  Both alternatives to ds_add_f32 were tried here: the cmpst loop lost (5.4 against 5.0 ms on
    // config 2: a drain iteration carries several items of one row; again in round 2 with the wave merges in place:
    // headline slab 3.73 -> 3.80 ms, config 2 4.31 -> 5.00 ms); 64-bit fixed-point rows with ds_add_u64 (28 B
    // per row, so 1728 rows in the workgroup's LDS share instead of 2048) came out mixed -- at equal row counts
    // (1024) 5.19 against 5.67 ms, but with 1728 rows bathroom 5.15 (float, 2048 rows: 4.96), pool 4.55 (5.01),
    // specular 22.0 (18.0), V = 7 829: 4.88 (4.31), V = 10^6: 5.52 (5.68) -- and was not kept.
    static constexpr uint32_t kStep = Acc::kBucketed ? 4u : 1u;
    __device__ __forceinline__ static uint32_t home(uint32_t key) {
        // Fixed-point tables: four consecutive keys share a BUCKET of four consecutive slots (probing moves by whole
        // buckets).  The flush walks the table in slot order, so the rows of neighbouring vertices -- a triangle's, its
        // neighbours' -- leave in the same wave instruction and, where they share a 64-byte line, as ONE atomic request
        // (the rate of scattered row atomics is 18.6 G/s, of rows that arrive line by line 108 G/s:
        // tools/micro/global_atomics.hip).  Pool caustic slab 4.04 -> 3.82 ms, config 5 0.155 -> 0.123 ms.  The float
        // table gains nothing from it (headline slab 3.73 -> 3.78 ms) and keeps one hash per key.
        static_assert(kTableSize % kStep == 0, "whole buckets");
        return Acc::kBucketed
            ? (uint32_t) (((unsigned long long) ((key >> 2) * 2654435761u) * (unsigned long long) (kTableSize / 4)) >> 32) * 4u + (key & 3u)
            : (uint32_t) (((unsigned long long) (key * 2654435761u) * (unsigned long long) kTableSize) >> 32);
    }
    __device__ __forceinline__ static uint32_t next(uint32_t slot) {
        return slot + kStep >= (uint32_t) kTableSize ? slot + kStep - (uint32_t) kTableSize : slot + kStep;
    }
    __device__ __forceinline__ void add_at(uint32_t slot, float x, float y, float z) const {
        if (Acc::kBucketed) {
            // integer rows: adding a zero costs an LDS operation, skipping it costs a branch (exec-mask bookkeeping) per
            // component in a loop whose run time is the instructions it issues
            Acc::add(&vals[3 * slot + 0], x); Acc::add(&vals[3 * slot + 1], y); Acc::add(&vals[3 * slot + 2], z);
            return;
        }
        if (x != 0.f) Acc::add(&vals[3 * slot + 0], x);
        if (y != 0.f) Acc::add(&vals[3 * slot + 1], y);
        if (z != 0.f) Acc::add(&vals[3 * slot + 2], z);
    }
    __device__ __forceinline__ void add(uint32_t key, float x, float y, float z) const {
        if (x == 0.f && y == 0.f && z == 0.f) return;
        if (!Acc::fits(x, y, z)) { global_add(key, x, y, z); return; }
        uint32_t slot = home(key);
        bool placed = false;
#pragma unroll 1
        for (int probe = 0; probe < kMaxProbe; ++probe) {
            const uint32_t prev = atomicCAS(&keys[slot], kEmptyKey, key);
            if (prev == kEmptyKey || prev == key) { placed = true; break; }
            slot = next(slot);
        }
        if (placed) add_at(slot, x, y, z);
#ifdef EPSM_KO_NOOVERFLOW                       // (knock-out build: what the rows that find no slot cost; results are wrong)
        else if (key == 0x12345678u) global_add(key, x, y, z);
#else
        else global_add(key, x, y, z);          // crowded neighbourhood: go straight to HBM (out of line it cost 11 spilled registers: 2.55 -> 2.70 ms)
#endif
    }
    // add() without the way out: false = the row found no slot (or does not fit a fixed-point row) and is still owed to the
    // buffers (drain_queue collects such rows and sends x, y, z of a row as ONE atomic request, global_add_component)
    __device__ __forceinline__ bool try_add(uint32_t key, float x, float y, float z) const {
        if (x == 0.f && y == 0.f && z == 0.f) return true;
        if (!Acc::fits(x, y, z)) return false;
        uint32_t slot = home(key);
        bool placed = false;
#pragma unroll 1
        for (int probe = 0; probe < kMaxProbe; ++probe) {
#ifdef EPSM_READ_BEFORE_CAS                // (A/B build: a plain read first -- most rows of a window are already there)
            uint32_t prev = keys[slot];
            if (prev == kEmptyKey) prev = atomicCAS(&keys[slot], kEmptyKey, key);
#else
            const uint32_t prev = atomicCAS(&keys[slot], kEmptyKey, key);
#endif
            if (prev == kEmptyKey || prev == key) { placed = true; break; }
            slot = next(slot);
        }
        if (placed) add_at(slot, x, y, z);
        return placed;
    }
    // component c (0..2) of a row straight to the buffers; alpha rows carry one component
    __device__ __forceinline__ void global_add_component(uint32_t key, int c, float v) const {
        if (!adds_something(v)) return;
        if (key < V) atomicAdd(gpos + 3 * (int64_t) key + c, v);
        else if (key < 2u * V) atomicAdd(gnrm + 3 * (int64_t) (key - V) + c, v);
        else if (c == 0) atomicAdd(galpha + (key - 2u * V), v);
    }
    // all threads of the workgroup; barriers inside
    __device__ __forceinline__ void clear() const {
        for (int e = threadIdx.x; e < kTableSize; e += blockDim.x) {
            keys[e] = kEmptyKey; vals[3 * e] = Val(0); vals[3 * e + 1] = Val(0); vals[3 * e + 2] = Val(0);
        }
        if (threadIdx.x == 0) *used = 0;
        __syncthreads();
    }
    // Lanes 4e..4e+2 add the x,y,z of row e in ONE wave instruction, so the three
    // components of a row (and neighbouring rows of a 64-B line) leave as one atomic
    // request instead of three (TCC_EA0_ATOMIC: -3x on the flush).
    // `last`: the workgroup's final flush -- the table is not used again, so it is not cleared (one window per workgroup: every flush)
    __device__ __forceinline__ void flush(bool last = false) const {
        __syncthreads();
        for (int q = threadIdx.x; q < 4 * kTableSize; q += blockDim.x) {
            const int e = q >> 2, c = q & 3;
            const uint32_t key = keys[e];
            if (key != kEmptyKey && c < 3) {
                const Val qv = vals[3 * e + c];
                if (qv != Val(0)) {
                    const float v = Acc::get(qv);
#ifdef EPSM_KO_NOFLUSHATOMICS
                    if (v == 1.2345e-30f)
#endif
                    if (key < V) atomicAdd(gpos + 3 * (int64_t) key + c, v);
                    else if (key < 2u * V) atomicAdd(gnrm + 3 * (int64_t) (key - V) + c, v);
                    else if (c == 0) atomicAdd(galpha + (key - 2u * V), v);
                }
            }
        }
        if (last) return;                                            // (workgroup-uniform)
        __syncthreads();
        for (int e = threadIdx.x; e < kTableSize; e += blockDim.x) {
            keys[e] = kEmptyKey; vals[3 * e] = Val(0); vals[3 * e + 1] = Val(0); vals[3 * e + 2] = Val(0);
        }
        if (threadIdx.x == 0) *used = 0;
        __syncthreads();
    }
    // Workgroup-wide census of occupied rows (all threads; barriers inside).  Counting at
    // chunk boundaries replaces a per-insertion counter, which was one hot LDS address.
    __device__ __forceinline__ bool crowded(int eighths = 4) const {
        __syncthreads();
        if (threadIdx.x == 0) *used = 0;
        __syncthreads();
        int c = 0;
        for (int e = threadIdx.x; e < kTableSize; e += blockDim.x) c += keys[e] != kEmptyKey;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
        if ((threadIdx.x & 63) == 0) atomicAdd(used, c);
        __syncthreads();
        // stand-alone scatter kernel: half full (3/8, 5/8, 6/8 measured: +0.2..0.4 ms on config 2); the fused kernel, whose
        // workgroups walk ADJACENT windows and so keep meeting the rows they already hold, waits until 6/8
        return *used > kTableSize * eighths / 8;
    }
};

// (Tried in round 2 and dropped: rows LOCKED with ds_cmpst_rtn -- top bit of the key -- and updated with plain loads,
// VALU adds and one 16-byte store that also releases the lock, i.e. no float atomics at all.  Correct (all tests), but
// 4.47 -> 5.02 ms on the headline slab and 4.92 -> 6.31 ms on config 2: the items of a drain iteration repeat their
// keys (adjacent lanes push the same triangles), every repeat is another round of three dependent LDS round trips,
// while the hardware float atomic resolves same-address lanes at ~2 clocks each.  The knock-out that drops the items
// instead of inserting them saves 0.55 / 0.88 ms: that is all there is to gain from the insertion.)

template <typename Table> __device__ __forceinline__ void atomic_add3(const Table &T, uint32_t key, V3<float> g) { T.add(key, g.x, g.y, g.z); }

// Sum over the wave with DPP adds only (no LDS crossbar): quad swaps, half-row and row mirrors
// give every lane its 16-lane row total, row_bcast15 / row_bcast31 chain the rows; lane 63 ends up
// with the wave total.  Lanes that must not contribute pass 0.  All 64 lanes must be active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_total_lane63(float v) {
    v = dpp_add<0xB1, 0xf>(v);     // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xf>(v);     // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xf>(v);    // row_half_mirror
    v = dpp_add<0x140, 0xf>(v);    // row_mirror
    v = dpp_add<0x142, 0xa>(v);    // row_bcast15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);    // row_bcast31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ float lane63(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }
__device__ __forceinline__ V3<float> wave_total_lane63(V3<float> v) {
    return mk3<float>(wave_total_lane63(v.x), wave_total_lane63(v.y), wave_total_lane63(v.z));
}

// Three rows of one merge round: lanes with `mine` are summed, the carrier lane receives the totals, the
// other merged lanes are zeroed.  Out of line: it is used ~20 times per kernel and the fused kernel has to
// stay inside the instruction cache.
struct Rows3 { float v[9]; };
#ifdef EPSM_MERGE_INLINE
__device__ __forceinline__ Rows3 merge_rows3(Rows3 r, bool mine, bool carrier) {
#else
__device__ __attribute__((noinline)) Rows3 merge_rows3(Rows3 r, bool mine, bool carrier) {
#endif
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        const float tot = lane63(wave_total_lane63(mine ? r.v[c] : 0.f));
        r.v[c] = carrier ? tot : (mine ? 0.f : r.v[c]);
    }
    return r;
}
#ifdef EPSM_MERGE_INLINE
__device__ __forceinline__ float merge_row1(float v, bool mine, bool carrier) {
#else
__device__ __attribute__((noinline)) float merge_row1(float v, bool mine, bool carrier) {
#endif
    const float tot = lane63(wave_total_lane63(mine ? v : 0.f));
    return carrier ? tot : (mine ? 0.f : v);
}

// fewer lanes than this: the DPP sums (VALU, the kernel's bottleneck) cost more than the LDS atomics they save
// (measured on the bathroom / specular / pool profiles: 4, 8, 16, 32 for the triangle rows)
constexpr int kMinMergeLanes = 16, kMinMergeLanesAlpha = 4;

// Lanes whose rows go to the SAME three parameter rows (the samples of one pixel at the first hit, the
// two triangles of an area light, one BSDF's alpha) are summed over the wave with DPP adds and the first
// of them alone carries the sum on: the LDS table then sees one row instead of up to 64 same-address
// atomics, which it executes one after the other.  Up to ROUNDS distinct targets per call; a round that
// would merge fewer than kMinMergeLanes lanes ends the search.
template <int ROWS, int ROUNDS>
__device__ __forceinline__ void merge_equal(bool &any, const uint32_t id[3], V3<float> vals[ROWS], int live_rows = ROWS) {
    constexpr int kMin = ROWS == 1 ? kMinMergeLanesAlpha : kMinMergeLanes;
#ifndef EPSM_CP_MERGE       // Round 4: OFF.  With integer LDS rows same-address lanes no longer serialise as float atomics did, and at three
    return;                 // waves per SIMD the DPP sums cost more issue slots than the table saves: 2.22 -> 2.17 ms without them.
#endif
    unsigned long long pending = __ballot(any);
#pragma unroll 1
    for (int round = 0; round < ROUNDS; ++round) {
        if (__popcll(pending) < kMin) return;
        const int leader = __ffsll((long long) pending) - 1;
        const uint32_t l0 = (uint32_t) __builtin_amdgcn_readlane((int) id[0], leader),
                       l1 = (uint32_t) __builtin_amdgcn_readlane((int) id[1], leader),
                       l2 = (uint32_t) __builtin_amdgcn_readlane((int) id[2], leader);
        const bool mine = any && id[0] == l0 && id[1] == l1 && id[2] == l2;
        const unsigned long long mm = __ballot(mine);
        pending &= ~mm;
        if (__popcll(mm) < kMin) return;                             // incoherent wave: stop searching
        const bool carrier = lane_id() == leader;
        if (ROWS == 1) {
            vals[0].x = merge_row1(vals[0].x, mine, carrier);        // alpha rows carry one component
        } else {
#pragma unroll
            for (int j = 0; j + 2 < ROWS; j += 3) {
                if (j >= live_rows) break;                           // wave-uniform: rows nobody has
                Rows3 r = {{vals[j].x, vals[j].y, vals[j].z, vals[j + 1].x, vals[j + 1].y, vals[j + 1].z,
                            vals[j + 2].x, vals[j + 2].y, vals[j + 2].z}};
                r = merge_rows3(r, mine, carrier);
                vals[j] = mk3<float>(r.v[0], r.v[1], r.v[2]);
                vals[j + 1] = mk3<float>(r.v[3], r.v[4], r.v[5]);
                vals[j + 2] = mk3<float>(r.v[6], r.v[7], r.v[8]);
            }
        }
        if (mine && !carrier) any = false;
    }
}

// Direct insertion: LDS atomics merge equal keys natively (same-address lanes serialise
// at LDS speed), which beats shuffle scans when runs are short.
template <typename Table> __device__ __forceinline__ void scatter_triangle_direct(const Table &buf, uint32_t base, bool valid,
                                                                                 const uint32_t key[3], const V3<float> val[3]) {
    if (valid) {
#pragma unroll
        for (int j = 0; j < 3; ++j) atomic_add3(buf, base + key[j], val[j]);
    }
}
// Adaptive: segmented scan only when the wave's runs are long enough to pay for the shuffles.
template <typename Table> __device__ __forceinline__ void scatter_triangle_adaptive(const Table &buf, uint32_t base, bool valid,
                                                                                   const uint32_t key[3], const V3<float> val[3],
                                                                                   int max_runs) {
    const unsigned long long vm = __ballot(valid);
    if (vm == 0ull) return;
    const Runs r = make_runs(valid, key[0], key[1], key[2]);
    const int n_tails = __popcll(__ballot(r.tail));
    if (n_tails > max_runs) { scatter_triangle_direct(buf, base, valid, key, val); return; }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const V3<float> tot = seg_sum3(valid ? val[j] : zero3<float>(), r.head);
        if (r.tail) atomic_add3(buf, base + key[j], tot);
    }
}


// ---------------------------------------------------------------------------
// Per-wave item queue in LDS.
//
// Inside the gradient kernel rows become available under heavy divergence (at vertex 3
// a few percent of the lanes hold a gradient), so inserting them into the table in place
// issues mostly empty LDS atomics and exposes their latency in a kernel that only has
// two waves per SIMD.  Instead every lane APPENDS its rows to a wave-private queue
// (position = wave-uniform count + prefix popcount of the ballot: plain ds_write_b128,
// nothing to wait for) and the wave drains the queue with all 64 lanes busy.
// ---------------------------------------------------------------------------
struct QItem { uint32_t key; float x, y, z; };

// Inlined: as an out-of-line function (which kept the kernel at 54 KB of code) every call began with the callee's
// s_waitcnt vmcnt(0), i.e. waited for the vertex records the path code had prefetched; with the larger queues there
// are few enough call sites taken that inlining wins 1.5-4.5 % (config 2 4.60 -> 4.54 ms, V = 10^6 5.68 -> 5.44).
// (Round 3: kU items per lane read together and their probes issued together -- the insertion is a chain of dependent
// LDS round trips -- measured on the constraint-parallel kernel, headline slab, ms: kU = 1 2.89, 2 2.75, 4 3.02, 8 3.46
// against 2.66 for the loop below: the drain is not waiting for its own latency.)
template <typename Table>
__device__ __forceinline__ void drain_queue(QItem *q, int n, Table T) {
    // q is LDS: say so.  Through the generic pointer of this out-of-line function the read was a FLAT load, whose
    // s_waitcnt vmcnt(0) also waits for every global load in flight -- the vertex records the path code had
    // prefetched -- at each drain.
    typedef __attribute__((address_space(3))) uint32_t LdsWord;
    LdsWord *ql = (LdsWord *) q;
    // (Two items per lane and turn, their compare-and-swaps in flight together -- a probe is an LDS round trip the lane
    // waits for -- was slower: headline slab 3.67 -> 4.16 ms, config 2 4.19 -> 4.73, pool caustic 3.79 -> 4.03.)
    // Rows that find no slot within kMaxProbe buckets (a window whose distinct rows outnumber the table: the deep vertices of
    // the caustic and specular profiles) used to leave from the lane that held them, three float atomics = three requests per
    // row.  Now the wave collects them -- compacted into the slots of the queue this turn has just read -- and sends them
    // FOUR LANES PER ROW, x, y, z side by side in one instruction, as the flush does: one request per row (knock-out of the
    // overflow rows: pool slab 2.59 -> 2.23 ms, dense specular slab 13.7 -> 6.0: that is what they cost).
    const int lane = lane_id();
#pragma unroll 1
    for (int base = 0; base < n; base += 64) {                       // wave-uniform
        const int idx = base + lane;
        bool owed = false;
        QItem it; it.key = 0u; it.x = it.y = it.z = 0.f;
        if (idx < n) {
            it.key = ql[4 * idx]; it.x = __uint_as_float(ql[4 * idx + 1]); it.y = __uint_as_float(ql[4 * idx + 2]); it.z = __uint_as_float(ql[4 * idx + 3]);
#ifndef EPSM_KO_NOINSERT                   // (knock-out build: what the insertion into the table costs)
            owed = !T.try_add(it.key, it.x, it.y, it.z);
#else
            if (it.key == 0x12345678u && it.x == 1.2345f) owed = !T.try_add(it.key, it.x, it.y, it.z);
#endif
        }
#ifdef EPSM_KO_NOOVERFLOW                  // (knock-out build: what the rows that find no slot cost; results are wrong)
        owed = owed && it.key == 0x12345678u;
#endif
        const unsigned long long m = __ballot(owed);
        if (m != 0ull) {                                             // wave-uniform, rare where the table holds the window's rows
            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned) (m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) m, 0u));
            if (owed) { LdsWord *d = ql + 4 * (base + rank); d[0] = it.key; d[1] = __float_as_uint(it.x); d[2] = __float_as_uint(it.y); d[3] = __float_as_uint(it.z); }
            const int n_owed = __popcll(m);
            __builtin_amdgcn_wave_barrier();                         // (LDS operations of one wave execute in order; this keeps the compiler from moving the reads up)
#pragma unroll 1
            for (int j0 = 0; j0 < n_owed; j0 += 16) {
                const int j = j0 + (lane >> 2), c = lane & 3;
                if (j < n_owed && c < 3) {
                    const uint32_t key = ql[4 * (base + j)];
                    T.global_add_component(key, c, __uint_as_float(ql[4 * (base + j) + 1 + c]));
                }
            }
        }
    }
}

template <int CAP>
struct WaveQueue {
    QItem *q;          // this wave's CAP slots
    int count;         // wave-uniform

    // Every valid lane appends ROWS consecutive items (one ballot for the whole group); the queue is drained first when THESE
    // items would not fit (round 3 drained whenever 64 lanes' worth might not: with a queue of one full group -- 192 items -- that
    // was before every group, and the drain's last iteration ran partly filled each time).
    template <int ROWS, typename Table>
    __device__ __forceinline__ void push_rows(const Table &T, bool valid, const uint32_t key[ROWS], const V3<float> val[ROWS]) {
        const unsigned long long m = __ballot(valid);
        if (m == 0ull) return;
        const int n = ROWS * __popcll(m);
#ifdef EPSM_QUEUE_RESERVE_WORST            // (A/B build)
        if (count + 64 * ROWS > CAP) drain(T);
#else
        if (count + n > CAP) drain(T);
#endif
        const int below = __builtin_amdgcn_mbcnt_hi((unsigned) (m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) m, 0u));
        if (valid) {
            QItem *dst = q + count + ROWS * below;
#pragma unroll
            for (int j = 0; j < ROWS; ++j) {
                QItem it; it.key = key[j]; it.x = val[j].x; it.y = val[j].y; it.z = val[j].z;
                dst[j] = it;
            }
        }
        count += n;
    }
    template <typename Table> __device__ __forceinline__ void drain(const Table &T) {
        if (count > 0) drain_queue(q, count, T);
        count = 0;
    }
    // Drains the full groups of 64 items -- every drain iteration with all lanes busy -- and keeps the remainder (< 64 items) at
    // the head of the queue: afterwards items [64, CAP) are free for the caller until the next push.
    template <typename Table> __device__ __forceinline__ void drain_full_groups(const Table &T) {
        const int n_full = count & ~63;
        if (n_full > 0) drain_queue(q + (count - n_full), n_full, T);
        count -= n_full;
    }
    // room for a group of `rows` rows from all 64 lanes?
    template <typename Table> __device__ __forceinline__ void reserve(const Table &T, int rows) {
        if (count + 64 * rows > CAP) drain(T);
    }
};

}  // namespace epsm
