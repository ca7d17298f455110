// epsm_backward_cp.hip -- the backward pass (first-vertex tangent + calc_grad + parameter scatter, epsm.py:238-297) in
// CONSTRAINT-PARALLEL form: one lane per (path, constraint vertex), arithmetic in epsm_cp_core.h.
//
// A persistent workgroup walks windows of consecutive paths.  Per window:
//   plan    every path's flag word -> which terms exist and how many constraint vertices m it has (cp::*_plan);
//   sort    the window's paths are counting-sorted by m (stable), so that a ROUND -- 64 lanes -- holds 64 / max(m,1)
//           paths of ONE m, each path on max(m,1) adjacent lanes (lane = first lane of the path + k - 1);
//   rounds  dealt round-robin to the waves, no barrier between them.  A lane loads vertex k and the geometry of its two
//           neighbours, evaluates its constraint(s) once (unit-seed Jacobians stay in registers), the block recursions
//           run across the path's lanes with shuffles, and every lane then emits the rows of ITS vertex: wave-level
//           merge of equal targets (DPP), append to the wave's LDS queue, LDS accumulator table, float atomics on
//           flush -- the scatter back end of epsm_grad_scatter.hip (epsm_wave_scatter.h), unchanged.
// Against the one-lane-per-path kernel: no per-vertex state (256 -> ~half the VGPRs), every constraint evaluated once
// instead of twice and its frame once instead of three times, all lanes busy whatever the chain length, vertex loops are
// run-time loops (one instantiation per variant instead of one per K, a fraction of the code).
#include <stdlib.h>
#include <string.h>

#define EPSM_FAST_RCP64                 // epsm_path_core.h: 1 / x in float64 by v_rcp_f64 + two Newton steps
// Round 5: ONE trip to memory per record (EPSM_CP_STASH, below) is the product build -- headline slab 1.97 -> 1.90 ms, config 2
// 2.30 -> 2.25, config 5 0.097 -> 0.092, pool slab 2.61 -> 2.62 (profiles/r05_e_whole_line_ab.txt); -DEPSM_CP_NO_STASH: round 4's
// second trip after the recursions.  EPSM_CP_DMA / EPSM_CP_DMA_ALIAS / EPSM_CP_COOP: the whole-line request forms, measured slower.
#if !defined(EPSM_CP_NO_STASH) && !defined(EPSM_CP_STASH)
#define EPSM_CP_STASH
#endif
// ... and the lane's own record of the wave's NEXT round is requested before this round's emission (EPSM_CP_PREFETCH = 1, below):
// headline slab 1.89 -> 1.81 ms, config 2 2.23 -> 2.19, pool slab 2.61 -> 2.59; = 2 (+ the rays) and = 3 (+ the end-point record)
// spill 13-16 more registers and are slower (1.89 / 2.09 ms).  -DEPSM_CP_NO_PREFETCH: loads at the start of the round.
#if defined(EPSM_CP_STASH) && !defined(EPSM_CP_NO_PREFETCH) && !defined(EPSM_CP_PREFETCH)
#define EPSM_CP_PREFETCH 1
#endif
#include "epsm_fused.h"
#include "epsm_cp_core.h"
#include "epsm_wave_scatter.h"

using namespace epsm;

namespace {

#ifndef EPSM_CP_THREADS
#define EPSM_CP_THREADS 256
#endif
// Shape (round 4): workgroups of four waves, THREE of them per CU = three waves per SIMD.  That takes <= 168 registers (the
// kernel has 168, no scratch access inside the round loop: csrc/build/asm, tools/cp_regs.sh) and <= 53 248 bytes of LDS per
// workgroup -- a CU hands out 159 744 bytes (2 x 79 872 fitted in round 3, 3 x 53 872 does not, 3 x 53 120 does): the wave
// queues went from 384 to 192 items (one push of three rows from every lane), the table from 1504 to 960 fixed-point rows.
// (384-thread workgroups, two per CU: the second one is never co-resident -- 1485 waves in flight of 3072, 3.40 ms; 768
// threads, one per CU: 2.53 ms -- twelve waves meet at every barrier of a window; 192 threads, four per CU: 2.83 ms.)
#ifndef EPSM_CP_OCC
#define EPSM_CP_OCC 3                   // waves per SIMD the register budget is set for
#endif
#ifndef EPSM_CP_ROWS_FLOAT
#define EPSM_CP_ROWS_FLOAT 2064
#endif
#ifndef EPSM_CP_ROWS_FIXED
#define EPSM_CP_ROWS_FIXED 1184
#endif
#ifndef EPSM_CP_QUEUE
#define EPSM_CP_QUEUE 192               // >= 3 rows x 64 lanes, the largest push
#endif
// One window per workgroup: the hardware's dispatcher balances the windows over the 768 resident workgroups better than
// contiguous ranges per persistent workgroup did (headline slab, ms: 768 workgroups 2.22, 1536 2.32, 2048 -- round 3's
// choice at 512 resident ones -- 2.14, 3072 2.16, 4096 2.12, 8192 = one window each 2.07); the table is flushed about once
// per window anyway.  A launch of more windows than this walks several per workgroup.
#ifndef EPSM_CP_BLOCKS
#define EPSM_CP_BLOCKS (1 << 20)
#endif
constexpr int kThreads = EPSM_CP_THREADS, kWaves = kThreads / 64, kQueueCap = EPSM_CP_QUEUE;
constexpr int kKeys = 6;                // m = 0..5
constexpr int kStashFirstItem = 64;     // EPSM_CP_STASH: queue items [64, 192) = 2 048 bytes hold 32 bytes per lane between a round's loads and its emission
static_assert(kQueueCap - kStashFirstItem >= 128, "the stash needs 128 queue items");

// lane -> (path slot j, vertex k) inside a round of c lanes per path: j = lane / c without a division
__device__ __forceinline__ int div_small(int lane, int c) {
    const int inv = c == 1 ? 65536 : c == 2 ? 32768 : c == 3 ? 21846 : c == 4 ? 16384 : 13108;
    return (lane * inv) >> 16;
}

// A lane's value of the lane below / above it.  __shfl_up / __shfl_down are ds_bpermute_b32 -- a trip through the LDS crossbar, ~50 of
// them per round with a constraint chain, each waited for inside the recursions' dependency chains, on the same LDS pipe as the other
// waves' table atomics; the whole-wave shifts of the data-parallel primitives (DPP wave_shr:1 / wave_shl:1, gfx9 only) do the same in
// ONE vector instruction without leaving the SIMD.  With `old` = the lane's own value and bound_ctrl off, lane 0 / lane 63 keep
// theirs, as __shfl_up / __shfl_down leave them.  (-DEPSM_CP_NO_DPP_SHIFT: the shuffles.)
#ifndef EPSM_CP_NO_DPP_SHIFT
__device__ __forceinline__ int up1_bits(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }      // wave_shr:1
__device__ __forceinline__ int down1_bits(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false); }    // wave_shl:1
__device__ __forceinline__ float up1(float v) { return __int_as_float(up1_bits(__float_as_int(v))); }
__device__ __forceinline__ float down1(float v) { return __int_as_float(down1_bits(__float_as_int(v))); }
__device__ __forceinline__ int up1(int v) { return up1_bits(v); }
__device__ __forceinline__ int down1(int v) { return down1_bits(v); }
__device__ __forceinline__ uint32_t up1(uint32_t v) { return (uint32_t) up1_bits((int) v); }
__device__ __forceinline__ uint32_t down1(uint32_t v) { return (uint32_t) down1_bits((int) v); }
#else
template <typename T> __device__ __forceinline__ T up1(T v) { return __shfl_up(v, 1); }
template <typename T> __device__ __forceinline__ T down1(T v) { return __shfl_down(v, 1); }
#endif
__device__ __forceinline__ V2<float> up1(V2<float> v) { return mk2<float>(up1(v.x), up1(v.y)); }
__device__ __forceinline__ M2<float> up1(M2<float> m) { M2<float> o; o.a = up1(m.a); o.b = up1(m.b); o.c = up1(m.c); o.d = up1(m.d); return o; }
__device__ __forceinline__ V3<float> down1(V3<float> v) { return mk3<float>(down1(v.x), down1(v.y), down1(v.z)); }
__device__ __forceinline__ V3<float> up1(V3<float> v) { return mk3<float>(up1(v.x), up1(v.y), up1(v.z)); }

// ---- record access on the reference's per-field arrays (the native log is read by geo_issue / addr_issue below); the flag
// word of a path from either
template <bool PACKED> struct Records {
    const FusedArgs &F;
    const PtrTable &P;                   // LDS (per-field arrays)
    int64_t i;
    __device__ __forceinline__ cp::Own<float> own(int k) const {
        cp::Own<float> o;
        const VertexPtrs<float> &v = P.v[k - 1];
        const Geo<float> g = load_geo(v, i);
        const Nrm<float> nr = load_nrm(v, i, g.b0, g.b1);
        o.x = g.x; o.e1 = g.e1; o.e2 = g.e2; o.b0 = g.b0; o.b1 = g.b1;
        o.n = nr.n; o.dn1 = nr.dn1; o.dn2 = nr.dn2; o.eta = lds_(v.eta, i); o.light = load3(v.light, i);
        return o;
    }
    __device__ __forceinline__ Geo<float> geo(int k) const { return load_geo(P.v[k - 1], i); }
    __device__ __forceinline__ V3<float> cam() const { return load3(F.g.cam, i); }
    __device__ __forceinline__ uint32_t tri_id(int k) const { return lds_(P.s[k - 1].tri, i); }
    __device__ __forceinline__ void b0b1(int k, float &b0, float &b1) const { b0 = lds_(P.v[k - 1].b0, i); b1 = lds_(P.v[k - 1].b1, i); }
    // emitter-sample record [etri, eb0, eb1, eweight]
    __device__ __forceinline__ bool emit(int k, uint32_t &etri, float &eb0, float &eb1, float &ew) const {
        const uint32_t *p = P.s[k - 1].emit;
        if (!p) return false;
        const U4 e4 = load_u4(p, i);
        etri = e4.x; eb0 = bits_to_float(e4.y); eb1 = bits_to_float(e4.z); ew = bits_to_float(e4.w);
        return true;
    }
    // BSDF record: alpha slot and d hf / d alpha
    __device__ __forceinline__ void aux(int k, uint32_t &bid, V3<float> &dhf) const {
        bid = kNoIndex; dhf = zero3<float>();
        if (!F.galpha || !P.s[k - 1].aux) return;
        const U4 a4 = load_u4(P.s[k - 1].aux, i);
        bid = a4.x; dhf = mk3<float>(bits_to_float(a4.y), bits_to_float(a4.z), bits_to_float(a4.w));
    }
    __device__ __forceinline__ uint32_t flag_word() const {
        if (PACKED) return lds_(F.pk_flags, i);
        uint32_t w = 0;
        for (int k = 1; k <= F.K; ++k) {
            const VertexPtrs<float> &v = P.v[k - 1];
            const uint32_t b = lds_(v.bsdf, i);
            w |= (((b & kBsdfDiffuse) ? 1u : 0u) | ((b & kBsdfNull) ? 2u : 0u) | (lds_(v.active, i) ? 4u : 0u) |
                  (lds_(v.active_em, i) ? 8u : 0u) | (lds_(v.ismesh, i) ? 16u : 0u)) << (5 * (k - 1));
        }
        return w;
    }
};

// ---- emission: the rows of one vertex into the wave queue (clamp / NaN rule of calc_grad per (N,3) component first,
// then the linear map of epsm_scatter_core.h: epsm.py:559-562, 622-627, 644-645).  One GROUP of three rows after the other
// is formed in one reused V3 vals[3], merged over the wave (DPP) and pushed -- positions (with the flat-normal part and
// diffuse_grad[0] folded in), normals, emitter, alpha, end point, occluder -- so that what a group needed is dead before
// the next one starts (round 3 held pos[3], nrm[3], vals[6] and the emitter / alpha rows together: part of what pinned the
// kernel at 256 registers).  All 64 lanes call it.
template <typename Table> struct Emitter {
    const FusedArgs &F;
    const Table &T;
    WaveQueue<kQueueCap> &Q;

    template <int ROWS>
    __device__ __forceinline__ void push(bool valid, const uint32_t key[ROWS], const V3<float> val[ROWS]) const {
        Q.template push_rows<ROWS>(T, valid, key, val);
    }
    __device__ __forceinline__ V3<float> fin(V3<float> g) const {
        return mk3<float>(finalize(g.x, F.g.clip), finalize(g.y, F.g.clip), finalize(g.z, F.g.clip));
    }
    __device__ __forceinline__ static bool tri_ok(const U4 &t, int64_t V) {
        return t.x < (uint64_t) V && t.y < (uint64_t) V && t.z < (uint64_t) V;
    }
    // three rows on keys (a, b, c), merged over the wave and queued
    __device__ __forceinline__ void rows3(bool v, V3<float> vals[3], uint32_t a, uint32_t b, uint32_t c) const {
        const uint32_t keys[3] = {a, b, c};
        merge_equal<3, 2>(v, keys, vals);
        push<3>(v, keys, vals);
    }
    // three rows g w_j
    __device__ __forceinline__ void group(bool v, V3<float> g, float w0, float w1, uint32_t a, uint32_t b, uint32_t c) const {
        if (__ballot(v) == 0ull) return;
        V3<float> vals[3] = {g * w0, g * w1, g * (1.f - w0 - w1)};
        rows3(v, vals, a, b, c);
    }
    // `on`: this lane has a vertex whose parameter rows exist; `live`: it has a constraint (its end-point rows exist);
    // `d1`: it carries diffuse_grad[0] = dldp of a path whose first hit is diffuse (epsm.py:791-792, 998-1000) and the
    // occluder term (609-620).  n, e1, e2, b0, b1: geometry of the lane's vertex (b0, b1 also for a d1 lane without constraint).
    __device__ __forceinline__ void vertex(bool on, bool live, bool d1, V3<float> Gx, V3<float> gn, V3<float> gm, V3<float> glight,
                                           V3<float> gdiff, V3<float> dp, V3<float> n, V3<float> e1, V3<float> e2, float b0, float b1,
                                           float nb0, float nb1, uint32_t bid, V3<float> dhf, float eb0, float eb1, float ew,
                                           const U4 &t, const U4 &tn, const U4 &er, const U4 &sh, const U4 &ts) const {
        const V3<float> z = zero3<float>();
        const bool idx_ok = tri_ok(t, F.V), pos_ok = idx_ok && (t.w & kModePos);
        gn = fin(gn);
        const bool gn_on = on && idx_ok && nz3(gn);
        const float sgn = (t.w & kModeFlip) ? -1.f : 1.f;
        const V3<float> dpf = d1 ? fin(dp) : z;                               // diffuse_grad[0]
        {   // ---- position rows of the hit triangle: si.p_j * path_grad[5it+j]  +  si_follow.p * diffuse_grad[0] (561-562)
            const float b2 = 1.f - b0 - b1;
            V3<float> P[3] = {on ? fin(Gx * b0) : z, on ? fin(Gx * b1) : z, on ? fin(Gx * b2) : z};
            if (gn_on && !(t.w & kModeVertexNormals) && pos_ok) {
                // flat: sh = sgn normalize(cross(p1-p0, p2-p0)) with p1-p0 = e2-e1, p2-p0 = -e1   (mesh.cpp:729, 811)
                const V3<float> d0 = e2 - e1, d1_ = -e1;
                const V3<float> cr = cross(d0, d1_);
                const float il = rsqrt_(dot(cr, cr));
                const V3<float> ch = cr * il;
                const V3<float> cb = (gn - ch * dot(ch, gn)) * (il * sgn);
                const V3<float> d0b = cross(d1_, cb), d1b = cross(cb, d0);
                P[1] = P[1] + d0b; P[2] = P[2] + d1b; P[0] = P[0] - (d0b + d1b);
            }
            if (d1) { P[0] = P[0] + dpf * b0; P[1] = P[1] + dpf * b1; P[2] = P[2] + dpf * b2; }
            const bool v = pos_ok && (nz3(P[0]) || nz3(P[1]) || nz3(P[2]));
            if (__ballot(v) != 0ull) rows3(v, P, t.x, t.y, t.z);
        }
        {   // ---- normal rows: si_follow.sh_frame.n * path_grad[5it+3]; logged normals are post-flip: n = sum_j b_j n'_j,
            // sh = normalize(n)   (mesh.cpp:784-790, 820-827)
            V3<float> pg = z;
            if (gn_on && (t.w & kModeVertexNormals) && (t.w & kModeNrm)) {
                const float il = rsqrt_(dot(n, n));
                const V3<float> sh_ = n * il;
                pg = (gn - sh_ * dot(sh_, gn)) * (il * sgn);
            }
            const uint32_t V32 = (uint32_t) F.V;
            group(nz3(pg), pg, b0, b1, V32 + t.x, V32 + t.y, V32 + t.z);
        }
        {   // ---- emitter rows: si_direct.p * light_grad[it] * sum(Lr_dir)  (epsm.py:622-627); area lights are a handful of triangles
            glight = fin(glight);
            const bool v = on && nz3(glight) && tri_ok(er, F.V) && (er.w & kModePos);
            group(v, glight * ew, eb0, eb1, er.x, er.y, er.z);
        }
        {   // ---- alpha row: bsdf_sample.hf * path_grad[5it+4]  (epsm.py:645); a handful of materials
            gm = fin(gm);
            bool v = on && nz3(gm) && bid < (uint64_t) F.B;
            if (__ballot(v) != 0ull) {
                V3<float> val[1] = {mk3<float>(v ? dot(gm, dhf) : 0.f, 0.f, 0.f)};
                const uint32_t aid[3] = {2u * (uint32_t) F.V + bid, 0u, 0u};
                merge_equal<1, 4>(v, aid, val);
                push<1>(v, aid, val);
            }
        }
        {   // ---- end-point rows: si_follow.p * diffuse_grad[it] with detached barycentrics (epsm.py:561-562), the NEXT vertex's triangle
            gdiff = fin(gdiff);
            const bool v = live && nz3(gdiff) && tri_ok(tn, F.V) && (tn.w & kModePos);
            group(v, gdiff, nb0, nb1, tn.x, tn.y, tn.z);
        }
        {   // ---- occluder of the first vertex's emitter sample: si_direct.p * diffuse_grad[0] * dis (epsm.py:609-620);
            // sh = [stri, sb0, sb1, dis] (EpsmScatterRecord.shadow), ts = the occluder triangle's row of the scene table
            const float dis = bits_to_float(sh.w);
            const bool v = d1 && nz3(dpf) && tri_ok(ts, F.V) && (ts.w & kModePos) && dis != 0.f;
            group(v, dpf * dis, bits_to_float(sh.y), bits_to_float(sh.z), ts.x, ts.y, ts.z);
        }
    }
};

// ---- who a lane is in round r of the current window
struct LaneId { int q, c, k; uint32_t plan; int loc; };      // plan == 0: no path on this lane
struct Rounds {
    int cls[kKeys + 1], rb[kKeys + 1];
    const uint16_t *perm;            // LDS: the window's paths sorted by m
    const uint32_t *plan;            // LDS
    __device__ __forceinline__ LaneId lane_of(int r, int lane) const {
        LaneId L;
        int q = 0;
#pragma unroll
        for (int t = 1; t < kKeys; ++t) if (r >= rb[t]) q = t;
        int cls_q = cls[0], n_q = cls[1] - cls[0], rb_q = rb[0];
#pragma unroll
        for (int t = 1; t < kKeys; ++t) if (q == t) { cls_q = cls[t]; n_q = cls[t + 1] - cls[t]; rb_q = rb[t]; }
        const int c = q > 0 ? q : 1, ppr = 64 / c;
        const int j = div_small(lane, c), k = lane - j * c + 1;
        const int idx = (r - rb_q) * ppr + j;
        const bool lane_on = r >= 0 && r < rb[kKeys] && j < ppr && idx < n_q;      // (r < 0: past the last round handed out)
        const int loc = lane_on ? (int) perm[cls_q + idx] : 0;
        L.q = q; L.c = c; L.k = k; L.loc = loc;
        L.plan = lane_on ? plan[loc] : 0u;               // (paths beyond the end of the wavefront have plan 0)
        return L;
    }
};

// ---- what a round reads of the native log (include/epsm.h, EpsmPackedLog: a record is one 128-byte line = two 64-byte sectors;
// a CU's vector L1 keeps ~64 sector misses in flight, so what a round costs is the sectors it touches times their latency):
//   geo     at the start of a round: the lane's own record -- quads 0..5 of a constraint vertex (both sectors), quads 0..2
//           (geometry, barycentrics, triangle id: the FIRST sector only) of a first hit that is diffuse without being a
//           constraint; the first lane of a path its rays (12 words) and the image gradient of its pixel; the LAST lane of a
//           path whose chain ends on a vertex without a lane of its own (the diffuse end point: k + 1 = nv > m) quads 0..2 of
//           record k + 1 -- one sector.  The geometry of vertices k - 1 and k + 1 that DO have a lane comes from that lane (one
//           shuffle per word) instead of being read again;
//   addr    after the recursions: the words only the EMISSION needs -- emitter sample (quad 6), d hf / d alpha (quad 7), the
//           occluder record; they and the rows of the scene table (requested after the second sweep: rows of triangles that
//           neighbouring paths hit too, mostly found in the vector cache) are NOT live across the recursions, where the 2x2
//           blocks in float64 need the registers.  (Rounds 3 and 4 prefetched -- into 62 registers, then by one-word touches;
//           the triangle's vertex rows INSIDE the record instead of the table lookup measured slower: MEASUREMENTS.md 9.9.)
typedef float F2v __attribute__((ext_vector_type(2)));
typedef float F3v __attribute__((ext_vector_type(3)));
struct GeoFetch {                    // (named fields, no arrays: the struct has to end up in registers, not in scratch memory)
    F4v o0, o1, o2, o3, o4, o5;      // own record, quads 0..5
    float o_lz;                      // ... word 28: light.z (constraint versions with an emitter sample)
    F4v p0, p1, p2;                  // first lane: the rays
    F4v n0, n1, n2;                  // last lane, record k+1: quads 0..2
    float gx, gy;
};
struct AddrFetch {
    F4v q6;                          // emitter sample [etri, eb0, eb1, eweight]
    F4v q7;                          // [light.z, d hf / d alpha]
    U4 sh;                           // the first vertex's occluder record (max_depth <= 3 logs)
};
// (Only words that are USED are loaded: a register of a pending load's destination that nobody reads is free for the
// register allocator, and the hardware's write to it then has to be waited for.)
__device__ __forceinline__ F2v ld2(const float *p) { return *(const __attribute__((address_space(1))) F2v *) p; }
// Window-uniform bases (scalar registers): a lane's addresses are these plus a 32-bit offset -- `global_load ... v_off, s[base]`
// instead of 64-bit multiply-adds per lane and load group -- and its pixel comes from the window's first pixel by two small
// divisions (float reciprocal + one correction step each way, exact below 2^24) instead of two 32-bit integer divisions.
struct WinBase {
    const float *verts, *rays;
    const uint32_t *shadow;
    const uint32_t *list;            // EpsmPackedLog.path_list + the window's first slot: path of slot loc = list[loc] (null: base + loc)
    int64_t base;
    uint32_t pix0, rem0;             // (path_offset + base) / spp and the remainder
    float rcp_spp, rcp_res;
    bool small;                      // pixel indices of this film stay below 2^24
};
__device__ __forceinline__ WinBase win_base(const FusedArgs &F, int64_t base) {
    WinBase B;
    B.base = base;
    B.verts = F.pk_verts + base * F.pk_path_stride;
    B.rays = F.pk_rays + base * F.pk_ray_stride;
    B.shadow = F.pk_shadow ? F.pk_shadow + 4 * base : nullptr;
    B.list = F.pk_list ? F.pk_list + base : nullptr;
    const int64_t p0 = F.tin.path_offset + base, q0 = p0 / F.tin.spp;
    B.small = (int64_t) F.tin.res * F.tin.res < (1 << 24) && q0 + 4096 < (1 << 24) && F.tin.spp < (1 << 20);
    B.pix0 = (uint32_t) q0; B.rem0 = (uint32_t) (p0 - q0 * F.tin.spp);
    B.rcp_spp = 1.f / (float) F.tin.spp; B.rcp_res = 1.f / (float) F.tin.res;
    return B;
}
// x / d and x % d for 0 <= x < 2^24, d >= 1 (float reciprocal + one correction step each way)
__device__ __forceinline__ void divmod24(uint32_t x, uint32_t d, float rcp_d, uint32_t &qo, uint32_t &ro) {
    uint32_t q = (uint32_t) ((float) x * rcp_d);
    int32_t r = (int32_t) x - (int32_t) (q * d);
    if (r < 0) { --q; r += (int32_t) d; }
    if (r >= (int32_t) d) { ++q; r -= (int32_t) d; }
    qo = q; ro = (uint32_t) r;
}
struct LaneRole { bool ok, first, live, end_next, d1, act1; uint32_t loc; const float *rec, *rays; const uint32_t *shadow; int64_t path; };
template <bool LIST>
__device__ __forceinline__ LaneRole role_of(const FusedArgs &F, const LaneId &L, const WinBase &B) {
    LaneRole R;
    R.ok = L.plan != 0u;
    R.first = L.k == 1;
    R.live = R.ok && L.q > 0;
    R.end_next = R.live && L.k == L.c && L.k + 1 <= cp::plan_nv(L.plan);      // vertex k+1 exists and has no lane
    R.d1 = R.ok && R.first && cp::plan_diffuse1(L.plan);
    R.act1 = (L.plan & cp::kPlanActive1) != 0;
    R.loc = (uint32_t) L.loc;
    if (LIST && B.list) {
        // the windows run over a LIST of paths (the tracer's survivors, EPSM_TRACE_FUSE_FIRST_HIT): slot -> path by one more load -- a
        // line the planning has just read -- and the addresses from the path itself
        const int64_t i = R.ok ? (int64_t) lds_(B.list, (int64_t) R.loc) : 0;
        R.path = i;
        R.rec = F.pk_verts + i * F.pk_path_stride + (L.k - 1) * kRecWords;
        R.rays = F.pk_rays + i * F.pk_ray_stride;
        R.shadow = F.pk_shadow ? F.pk_shadow + 4 * i : nullptr;
        return R;
    }
    R.path = -1;
    R.rec = B.verts + (uint32_t) (R.loc * (uint32_t) F.pk_path_stride + (uint32_t) (L.k - 1) * (uint32_t) kRecWords);
    R.rays = B.rays + (uint32_t) F.pk_ray_stride * R.loc;
    R.shadow = B.shadow ? B.shadow + 4 * R.loc : nullptr;
    return R;
}
// pixel of path i: (path_offset + i) / spp, row-major on the res x res crop (epsm.py:250)
// (films of 2^24 pixels and more: two 64-bit divisions, ~200 instructions that the round loop carried inline without ever
// running them -- out of line: headline slab 2.07 -> 2.06 ms, pool slab 2.79 -> 2.70)
__device__ __attribute__((noinline)) int64_t pixel_offset_large(int64_t path, int spp, int res, int img_width) {
    const int64_t pix = path / spp, y = pix / res;
    return y * img_width + (pix - y * res);
}
__device__ __forceinline__ const float *pixel_grad(const TangentIn &A, const WinBase &B, uint32_t loc, int64_t path = -1) {
    if (path >= 0) return A.grad_img + pixel_offset_large(A.path_offset + path, A.spp, A.res, A.img_width) * A.img_channels + 3;
    if (!B.small) return A.grad_img + pixel_offset_large(A.path_offset + B.base + loc, A.spp, A.res, A.img_width) * A.img_channels + 3;
    int64_t y, x;
    {
        uint32_t q1, r1, yy, xx;
        divmod24(B.rem0 + loc, (uint32_t) A.spp, B.rcp_spp, q1, r1);
        divmod24(B.pix0 + q1, (uint32_t) A.res, B.rcp_res, yy, xx);
        y = yy; x = xx;
    }
    return A.grad_img + (y * A.img_width + x) * A.img_channels + 3;
}
template <int VARIANT, bool LIST>
__device__ __forceinline__ void geo_issue(GeoFetch &X, const FusedArgs &F, const LaneId &L, const WinBase &B) {
    const LaneRole R = role_of<LIST>(F, L, B);
    // (a path WITHOUT a constraint reads its first record only when its first hit is diffuse: diffuse_grad[0] = dldp needs the
    // triangle; otherwise all it gives is its share of d/d ray.o, which needs the rays alone -- 27 % of the bathroom paths)
    if (R.live || R.d1) { X.o0 = ldq(R.rec, 0); X.o1 = ldq(R.rec, 1); X.o2 = ldq(R.rec, 2); }
    if (R.live) {
        X.o3 = ldq(R.rec, 3); X.o4 = ldq(R.rec, 4); X.o5 = ldq(R.rec, 5);
#ifndef EPSM_CP_STASH                   // (stash build: light.z arrives with quad 7)
        if (VARIANT == EPSM_VARIANT_MANIFOLD && cp::plan_a(L.plan, L.k)) X.o_lz = lds_(R.rec, 28);
#endif
    }
    if (R.ok && R.first) {
        const float *rays = R.rays;
#ifdef EPSM_CPKO_NORAYS                 // (knock-out build: what reading the rays costs; results are wrong)
        { const float v = (float) R.loc * 1e-3f; const F4v f4 = {v, 0.5f, -0.25f, 1.f}; X.p0 = f4; X.p1 = f4 * 0.5f; X.p2 = f4 * 0.25f; }
#elif defined(EPSM_CPKO_RAYS0)          // (knock-out build: every path takes the rays of its window's first path -- realistic values, no gather)
        X.p0 = ldq(B.rays, 0); X.p1 = ldq(B.rays, 1); X.p2 = ldq(B.rays, 2); (void) rays;
#else
        X.p0 = ldq(rays, 0); X.p1 = ldq(rays, 1); X.p2 = ldq(rays, 2);
#endif
        const F2v g = ld2(pixel_grad(F.tin, B, R.loc, R.path));
        X.gx = g.x; X.gy = g.y;
    }
    if (R.end_next) {
        const float *nx = R.rec + kRecWords;
        X.n0 = ldq(nx, 0); X.n1 = ldq(nx, 1); X.n2 = ldq(nx, 2);
    }
}
// ---- EPSM_CP_DMA (round 5): the lane's OWN record through LDS, as whole 128-byte lines.
// Per-lane loads ask the vector L1 for one 16-byte piece of 64 DIFFERENT lines per instruction -- six instructions per record,
// plus two more after the recursions for the words the emission needs: the same gathers with this kernel's amount of work
// beside them run at 3.6-3.9 TB/s (tools/micro/batch_gather.hip, profiles/r05_d_batch_gather.txt), which IS this kernel's
// measured HBM rate.  Eight lanes x 16 bytes per record in ONE instruction (LDS-DMA, global_load_lds_dwordx4: no register
// destination) make every request a whole line: 5.5-5.9 TB/s in the same micro-benchmark at the same three waves per SIMD,
// in BATCHES of 16 records through 2 304 bytes of staging per wave -- two DMA instructions, wait, the 16 owning lanes read
// their quads back, next batch.  Lane l of instruction j of batch b serves the record of lane `owner` = 16 b + 8 j + l / 8 and
// fetches its quad (l % 8) ^ (owner % 8) -- the XOR on the SOURCE side, the LDS image being lane-linear -- so that the 16
// owners' ds_read_b128 of one quad hit 16 different bank groups (slots 1 152 bytes apart: 1 024 + 128 of padding).
constexpr int kDmaSlotWords = 288, kDmaBatch = 16;
constexpr int kDmaStageWords = 2 * kDmaSlotWords;           // per wave
typedef __attribute__((address_space(3))) const F4v LdsF4;
__device__ __forceinline__ void dma16(const float *src, float *lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) src, (__attribute__((address_space(3))) void *) lds_dst, 16, 0, 0);
}
// want: 2 = the whole record (a constraint vertex), 1 = its first sector (a diffuse first hit without a constraint), 0 = nothing
template <int VARIANT, bool LIST>
__device__ __forceinline__ void geo_stage_dma(GeoFetch &X, const FusedArgs &F, const LaneId &L, const WinBase &B, float *stage, int lane) {
    const LaneRole R = role_of<LIST>(F, L, B);
    const uint32_t want = R.live ? 2u : R.d1 ? 1u : 0u;
    const uint32_t roff = (uint32_t) (R.loc * (uint32_t) F.pk_path_stride + (uint32_t) (L.k - 1) * (uint32_t) kRecWords) | want;   // (a multiple of 32: the low bits are free)
    const int s = lane & 7, grp = lane >> 3;
    const int quad = s ^ (grp & 7);                          // owner % 8 == (l / 8) % 8 in every instruction
    const bool want_lz = VARIANT == EPSM_VARIANT_MANIFOLD && R.live && cp::plan_a(L.plan, L.k);
#pragma unroll 1
    for (int b = 0; b < 64 / kDmaBatch; ++b) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t o = (uint32_t) __shfl((int) roff, kDmaBatch * b + 8 * j + grp);
            const uint32_t w = o & 3u;
            if (w == 2u || (w == 1u && quad < 4)) dma16(B.verts + (o & ~31u) + 4 * quad, stage + j * kDmaSlotWords);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int li = lane - kDmaBatch * b;
        if (li >= 0 && li < kDmaBatch && want != 0u) {
            const float *p = stage + (li >> 3) * kDmaSlotWords + s * kRecWords;
            X.o0 = *(LdsF4 *) (p + 4 * (0 ^ s)); X.o1 = *(LdsF4 *) (p + 4 * (1 ^ s)); X.o2 = *(LdsF4 *) (p + 4 * (2 ^ s));
            if (want == 2u) {
                X.o3 = *(LdsF4 *) (p + 4 * (3 ^ s)); X.o4 = *(LdsF4 *) (p + 4 * (4 ^ s)); X.o5 = *(LdsF4 *) (p + 4 * (5 ^ s));
                if (want_lz) X.o_lz = *(__attribute__((address_space(3))) const float *) (p + 4 * (7 ^ s));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the next batch overwrites the slots
    }
}
// ---- EPSM_CP_COOP: the same whole-line requests with REGISTER staging -- all eight cooperative loads of a round in flight at
// once (32 registers that nothing else needs at the start of a round), ONE wait, then the transposition through the 2 304-byte
// staging area in four batches of LDS round trips (~100 cycles each) instead of four trips to memory.
typedef __attribute__((address_space(3))) F4v LdsF4w;
template <int VARIANT, bool LIST>
__device__ __forceinline__ void geo_stage_coop(GeoFetch &X, const FusedArgs &F, const LaneId &L, const WinBase &B, float *stage, int lane) {
    const LaneRole R = role_of<LIST>(F, L, B);
    const uint32_t want = R.live ? 2u : R.d1 ? 1u : 0u;
    const uint32_t roff = (uint32_t) (R.loc * (uint32_t) F.pk_path_stride + (uint32_t) (L.k - 1) * (uint32_t) kRecWords) | want;
    const int s = lane & 7, grp = lane >> 3;
    const bool want_lz = VARIANT == EPSM_VARIANT_MANIFOLD && R.live && cp::plan_a(L.plan, L.k);
    const F4v z4 = {0.f, 0.f, 0.f, 0.f};
    F4v c0 = z4, c1 = z4, c2 = z4, c3 = z4, c4 = z4, c5 = z4, c6 = z4, c7 = z4;
    // instruction j: lane l fetches quad l % 8 of the record of lane 8 j + l / 8 (quads 4..7 only of whole records)
#define EPSM_COOP_LOAD(J, C) { const uint32_t o = (uint32_t) __shfl((int) roff, 8 * J + grp); const uint32_t w = o & 3u; \
        if (w == 2u || (w == 1u && s < 4)) C = ldq(B.verts + (o & ~31u), s); }
    EPSM_COOP_LOAD(0, c0) EPSM_COOP_LOAD(1, c1) EPSM_COOP_LOAD(2, c2) EPSM_COOP_LOAD(3, c3)
    EPSM_COOP_LOAD(4, c4) EPSM_COOP_LOAD(5, c5) EPSM_COOP_LOAD(6, c6) EPSM_COOP_LOAD(7, c7)
#undef EPSM_COOP_LOAD
    // transposition: instruction j's lane l holds quad s of local record grp; it goes to slot (j & 1), record grp, position s ^ grp
    float *const wr0 = stage + grp * kRecWords + 4 * (s ^ grp), *const wr1 = wr0 + kDmaSlotWords;
#define EPSM_COOP_BATCH(BATCH, CA, CB) { \
        *(LdsF4w *) wr0 = CA; *(LdsF4w *) wr1 = CB; \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        const int li = lane - kDmaBatch * BATCH; \
        if (li >= 0 && li < kDmaBatch && want != 0u) { \
            const float *p = stage + (li >> 3) * kDmaSlotWords + s * kRecWords; \
            X.o0 = *(LdsF4 *) (p + 4 * (0 ^ s)); X.o1 = *(LdsF4 *) (p + 4 * (1 ^ s)); X.o2 = *(LdsF4 *) (p + 4 * (2 ^ s)); \
            if (want == 2u) { \
                X.o3 = *(LdsF4 *) (p + 4 * (3 ^ s)); X.o4 = *(LdsF4 *) (p + 4 * (4 ^ s)); X.o5 = *(LdsF4 *) (p + 4 * (5 ^ s)); \
                if (want_lz) X.o_lz = *(__attribute__((address_space(3))) const float *) (p + 4 * (7 ^ s)); \
            } \
        } \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    EPSM_COOP_BATCH(0, c0, c1) EPSM_COOP_BATCH(1, c2, c3) EPSM_COOP_BATCH(2, c4, c5) EPSM_COOP_BATCH(3, c6, c7)
#undef EPSM_COOP_BATCH
}
// what stays a per-lane load beside the staged records: the rays + image gradient of a path's first lane, the first sector of
// the end-point record of its last lane
template <bool LIST>
__device__ __forceinline__ void geo_issue_rest(GeoFetch &X, const FusedArgs &F, const LaneId &L, const WinBase &B) {
    const LaneRole R = role_of<LIST>(F, L, B);
    if (R.ok && R.first) {
        const float *rays = R.rays;
        X.p0 = ldq(rays, 0); X.p1 = ldq(rays, 1); X.p2 = ldq(rays, 2);
        const F2v g = ld2(pixel_grad(F.tin, B, R.loc, R.path));
        X.gx = g.x; X.gy = g.y;
    }
    if (R.end_next) {
        const float *nx = R.rec + kRecWords;
#ifdef EPSM_CPKO_NOEND                  // (knock-out build: what reading the end-point records costs; results are wrong)
        X.n0 = X.o0 * 1.5f; X.n1 = X.o1 * 1.5f; X.n2 = X.o2; (void) nx;
#else
        X.n0 = ldq(nx, 0); X.n1 = ldq(nx, 1); X.n2 = ldq(nx, 2);
#endif
    }
}
template <int VARIANT, bool LIST>
__device__ __forceinline__ void addr_issue(AddrFetch &A, const FusedArgs &F, const LaneId &L, const WinBase &B) {
    const LaneRole R = role_of<LIST>(F, L, B);
    if (R.live) {
        const bool wN = VARIANT == EPSM_VARIANT_MANIFOLD && cp::plan_a(L.plan, L.k);
#ifdef EPSM_CP_STASH
        if (F.galpha || wN) A.q7 = ldq(R.rec, 7);                  // d hf / d alpha, and light.z of the emitter sample
#else
        if (F.galpha) A.q7 = ldq(R.rec, 7);
#endif
        if (wN) A.q6 = ldq(R.rec, 6);
    }
    if (R.d1 && R.shadow) A.sh = load_u4(R.shadow, 0);
}

// ---- EPSM_CP_PREFETCH (round 5; the product build): the lane's OWN record of the wave's NEXT round -- quads 0..5 and the
// emission's words -- requested before this round's emission, into 36 registers that only the emission has to live with (its
// pressure is ~80 registers below the recursions' peak: no additional spill), so that the trip to memory runs under the LDS work
// of the emission.  (Round 3 prefetched everything a round reads into 62 registers and paid for it with the third wave per SIMD.)
template <int VARIANT, bool LIST>
__device__ __forceinline__ void own_issue(GeoFetch &X, AddrFetch &A, const FusedArgs &F, const LaneId &L, const WinBase &B) {
    const LaneRole R = role_of<LIST>(F, L, B);
    if (R.live || R.d1) { X.o0 = ldq(R.rec, 0); X.o1 = ldq(R.rec, 1); X.o2 = ldq(R.rec, 2); }
    if (R.live) { X.o3 = ldq(R.rec, 3); X.o4 = ldq(R.rec, 4); X.o5 = ldq(R.rec, 5); }
    addr_issue<VARIANT, LIST>(A, F, L, B);
#if EPSM_CP_PREFETCH >= 2            // ... and the rays + image gradient of a path's first lane
    if (R.ok && R.first) {
        const float *rays = R.rays;
        X.p0 = ldq(rays, 0); X.p1 = ldq(rays, 1); X.p2 = ldq(rays, 2);
        const F2v g = ld2(pixel_grad(F.tin, B, R.loc, R.path));
        X.gx = g.x; X.gy = g.y;
    }
#endif
#if EPSM_CP_PREFETCH >= 3            // ... and the end-point record of its last lane
    if (R.end_next) {
        const float *nx = R.rec + kRecWords;
        X.n0 = ldq(nx, 0); X.n1 = ldq(nx, 1); X.n2 = ldq(nx, 2);
    }
#endif
}
__device__ __forceinline__ void fetch_zero(GeoFetch &X, AddrFetch &A) {
    const F4v z4 = {0.f, 0.f, 0.f, 0.f};
    X.o0 = X.o1 = X.o2 = X.o3 = X.o4 = X.o5 = X.p0 = X.p1 = X.p2 = X.n0 = X.n1 = X.n2 = z4;
    X.o_lz = X.gx = X.gy = 0.f;
    A.q6 = A.q7 = z4; A.sh.x = kNoIndex; A.sh.y = A.sh.z = A.sh.w = 0u;
}

// kWindow: the largest window (paths a workgroup plans, sorts and works through at a time); `window` <= kWindow, a multiple
// of 64, is what this launch uses -- a small wavefront is cut into smaller windows so that every CU gets some (launch()).
// Large wavefronts take windows of 2048 paths: every class of paths ends in a partly filled round, six of the 23 rounds of
// a 1024-path window of the bathroom profile and six of 42 at 2048 -- headline slab 2.48 -> 2.28 ms, config 2 2.82 -> 2.56,
// pool slab 3.54 -> 3.31 (3072: 2.39 / 2.82 / 3.62 and 4096: 2.12 / 2.78 / 3.46 -- the 6 bytes of LDS per path are taken from
// the table; with ONE partly filled round per window -- paths on 8 / 4 / 2 / 1 lanes, widest first, classes mixed inside a
// round -- 2.56: the mixed rounds run the longest chain's recursion for everybody).  Small wavefronts keep <= 1024.
#ifndef EPSM_CP_WINDOW
#define EPSM_CP_WINDOW 2048
#endif

// DROP: paths without any term take no lane (see kSortKeys below)
template <int VARIANT, int DMODE, bool PACKED, bool FLOAT_ROWS, int kWindow, bool DROP>
__global__ __launch_bounds__(kThreads, EPSM_CP_OCC) void epsm_backward_cp_kernel(FusedArgs F, int dcols, int64_t windows_per_block, int window) {
    // float rows where the window's distinct rows need the larger table (epsm_wave_scatter.h, AccFixed64)
    // Rows of the accumulator table: 64-bit fixed point (epsm_wave_scatter.h, AccFixed64: the LDS integer atomic inserts an
    // order of magnitude faster than ds_add_f32, and same-address lanes do not serialise as badly) unless the caller disabled
    // the outlier clamp -- then the terms are unbounded and the sums stay in float.
    constexpr bool kFloatRows = FLOAT_ROWS;
    // (a window costs 6 bytes of LDS per path: the larger one leaves 224 fixed-point / 384 float rows fewer)
    constexpr int kRowsFixed = kWindow > 1024 ? EPSM_CP_ROWS_FIXED - 224 * ((kWindow - 1024) / 1024) : EPSM_CP_ROWS_FIXED;
    constexpr int kRowsFloat = kWindow > 1024 ? EPSM_CP_ROWS_FLOAT - 384 * ((kWindow - 1024) / 1024) : EPSM_CP_ROWS_FLOAT;
    typedef LdsTable<kFloatRows ? kRowsFloat : kRowsFixed, typename std::conditional<kFloatRows, AccFloat, AccFixed64>::type> Table;
    constexpr int kTableSize = Table::kTableSize;
    __shared__ uint32_t s_keys[kTableSize];
    __shared__ typename Table::Val s_vals[kTableSize * 3];
    __shared__ int s_used;
    __shared__ QItem s_queue[kWaves][kQueueCap];
    __shared__ PtrTable s_ptrs;
#if defined(EPSM_CP_DMA) && !defined(EPSM_CP_DMA_ALIAS)
    __shared__ __attribute__((aligned(16))) float s_stage[PACKED ? kWaves : 1][PACKED ? kDmaStageWords : 4];
#endif
    float *const my_rep = F.rep ? F.rep + (blockIdx.x % (unsigned) F.replicas) * F.rep_stride : nullptr;
    const Table T{s_keys, s_vals, &s_used, my_rep ? my_rep : F.gpos, my_rep ? my_rep + 3 * F.V : F.gnrm,
                  my_rep ? my_rep + 6 * F.V : F.galpha, (uint32_t) F.V};
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    WaveQueue<kQueueCap> Q{s_queue[wv], 0};
    const Emitter<Table> E{F, T, Q};
    if (!PACKED && threadIdx.x < F.K) { s_ptrs.v[threadIdx.x] = F.g.v[threadIdx.x]; s_ptrs.s[threadIdx.x] = F.s[threadIdx.x]; }
    constexpr int kPer = (kWindow + kThreads - 1) / kThreads;        // paths a thread plans
    // classes the window is sorted into: the kKeys classes of the rounds; with DROP one more, the paths without a term, sorted behind
    // the others and handed to no round.  DROP is set with the caller's tangents, and with in-kernel tangents when the caller did not
    // ask for d loss / d ray.o = -sum grad_d (grad_o_sum == NULL): the reference forms that sum only `if dr.grad_enabled(ray.o)`
    // (epsm.py:258-259), i.e. when the sensor is being optimised -- EPSM/exp/bedroom.py, not bathroom.py and the others -- and without
    // it such a path (27 % of the bathroom profile) has nothing to give: no lane, no rays read.  Otherwise it gives its share of the
    // sum on a lane of class 0.  (A template parameter: as a run-time switch it cost the camera-gradient form 9 %, 1.78 -> 1.95 ms.)
    constexpr int kSortKeys = DROP ? kKeys + 1 : kKeys;
    // kList: the windows may run over a LIST of paths (EpsmPackedLog.path_list: the tracer's survivors under EPSM_TRACE_FUSE_FIRST_HIT --
    // 12 % of a traced wavefront, half of them with a term) instead of over all N: slot p of the launch is path list[p]
    constexpr bool kList = DROP && PACKED;
    constexpr int kStride = kPer * kWaves, kEntries = kSortKeys * kStride;
    __shared__ uint32_t s_plan[kWindow];
    __shared__ uint16_t s_perm[kWindow];
    __shared__ int s_cnt[kEntries];                                  // [m][j][wave]: histogram, then offsets
    __shared__ int s_cls[kKeys + 2];                                 // first sorted position of class m; [kKeys]: of the paths without a term
    V3<float> gd_acc = zero3<float>();                               // kTangentsInKernel: sum of grad_d over this lane's paths
    const int64_t n_slots = (kList && F.pk_list) ? (int64_t) lds_(F.pk_list_count, (int64_t) 0) : F.g.N;      // paths the windows run over
    const int64_t n_windows = (n_slots + window - 1) / window;
    if (kList && (int64_t) blockIdx.x * windows_per_block >= n_windows) return;      // (workgroup-uniform: a launch sized for all N)
    T.clear();                                                       // ends with a barrier: the table of pointers is visible too
#pragma unroll 1
    for (int64_t wi = 0; wi < windows_per_block; ++wi) {
        // a workgroup walks a CONTIGUOUS range of windows: neighbouring pixels keep hitting the rows its table holds
        const int64_t win = (int64_t) blockIdx.x * windows_per_block + wi;
        if (win >= n_windows) break;                                 // workgroup-uniform
        const int64_t base = win * window;
        const WinBase WB = win_base(F, base);
        // ---- plan + histogram of m: thread t plans paths base + j*kThreads + t.  With the caller's tangents a path WITHOUT ANY
        // TERM -- no constraint vertex, first hit not diffuse: 27 % of the bathroom paths -- gets key kKeys, a class sorted behind
        // the others and handed to no round.  (With in-kernel tangents such a path still owes its share of d/d ray.o = -sum grad_d,
        // epsm.py:260-261; taking that here, from coalesced loads of the rays, and keeping the path out of the rounds measured
        // 2.07 -> 2.15 ms on the headline slab: the planning runs before a barrier, a class-0 round under the other waves.)
        int key[kPer], rank[kPer];
        {
            uint32_t fw[kPer];
#pragma unroll
            for (int j = 0; j < kPer; ++j) {                         // all flag loads first
                int64_t p = base + j * kThreads + threadIdx.x;
                if (kList && F.pk_list) p = p < n_slots ? (int64_t) lds_(F.pk_list, p) : -1;      // slot -> path
                const Records<PACKED> R{F, s_ptrs, (p >= 0 && p < F.g.N) ? p : F.g.N - 1};
                fw[j] = (j * kThreads + (int) threadIdx.x < window && p >= 0 && p < F.g.N) ? R.flag_word() : 0u;
            }
#pragma unroll
            for (int j = 0; j < kPer; ++j) {
                const int loc = j * kThreads + threadIdx.x;
                const int64_t p = base + loc;
                const bool in = loc < window && p < n_slots;
                uint32_t w = fw[j];
                if (F.K < 5) w &= (1u << (5 * F.K)) - 1u;
                uint32_t plan = VARIANT == EPSM_VARIANT_MANIFOLD ? cp::manifold_plan(w) : cp::caustic_plan(w);
                plan = in ? (plan | cp::kPlanInRange | ((w & 4u) ? cp::kPlanActive1 : 0u)) : 0u;
                const bool none = kSortKeys > kKeys && (!in || (cp::plan_m(plan) == 0 && !cp::plan_diffuse1(plan)));
                key[j] = none ? kKeys : cp::plan_m(plan);
                if (loc < window) s_plan[loc] = plan;
            }
        }
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const bool has = j * kThreads + (int) threadIdx.x < window;
#pragma unroll
            for (int q = 0; q < kSortKeys; ++q) {
                const unsigned long long m = __ballot(has && key[j] == q);
                if (key[j] == q) rank[j] = __builtin_amdgcn_mbcnt_hi((unsigned) (m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) m, 0u));
                if (lane == 0) s_cnt[(q * kPer + j) * kWaves + wv] = __popcll(m);
            }
        }
        __syncthreads();
        if (wv == 0) {
            // exclusive scan in (m, j, wave) order: the whole window sorted by m, stable
            int carry = 0;
#pragma unroll
            for (int q0 = 0; q0 < kEntries; q0 += 64) {
                const int q = q0 + lane;
                const int c = q < kEntries ? s_cnt[q] : 0;
                int inc = c;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(inc, off); if (lane >= off) inc += t; }
                if (q < kEntries) {
                    s_cnt[q] = carry + inc - c;
                    if (q % kStride == 0) s_cls[q / kStride] = carry + inc - c;
                }
                carry += __shfl(inc, 63);
            }
            if (lane == 0) s_cls[kSortKeys] = window;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const int loc = j * kThreads + threadIdx.x;
            if (loc < window) s_perm[s_cnt[(key[j] * kPer + j) * kWaves + wv] + rank[j]] = (uint16_t) loc;
        }
        __syncthreads();
        // ---- rounds: class q has n_q paths on c = max(q,1) lanes each, 64 / c paths per round
        Rounds RS;
        RS.perm = s_perm; RS.plan = s_plan;
#pragma unroll
        for (int q = 0; q <= kKeys; ++q) RS.cls[q] = __builtin_amdgcn_readfirstlane(s_cls[q]);
        RS.rb[0] = 0;
#pragma unroll
        for (int q = 0; q < kKeys; ++q) {
            const int c = q > 0 ? q : 1, ppr = 64 / c;
            RS.rb[q + 1] = RS.rb[q] + (RS.cls[q + 1] - RS.cls[q] + ppr - 1) / ppr;
        }
#ifdef EPSM_CPKO_NOROUNDS                 // (knock-out builds, tools/build_cp_variant.sh: what a stage costs)
        RS.rb[kKeys] = 0;
#endif
        // Software pipeline: the records of the wave's NEXT round are requested before this round's rows go into the
        // queue / table, so the HBM round trip of one round runs under the LDS work of the other (measured apart they
        // were 1.6 ms and 1.5 ms per 2^24-path slab, and their sum when a wave did one after the other).
        // (Round r goes to wave r mod 4.  Handing the rounds out dynamically -- a wave takes the next one nobody has, one LDS
        // atomic per round -- measured 2.15 against 2.08 ms; the classes with the longest chains first, so that a window's tail
        // is made of cheap rounds: 6.3 ms, three times slower -- 9.8 ms in round 5, -DEPSM_CP_REVERSE_ROUNDS.  The reason: the ORDER fills the table.  The light
        // rounds come first and bring the rows a window shares -- the first-hit triangles of its pixels -- into their home slots; the deep vertices' rows,
        // which nobody shares, come last and leave for the buffers directly when their neighbourhood is full.  Reversed, the unshared rows take the slots
        // and the SHARED ones overflow: thousands of float atomics on the same few addresses.)
        const int n_rounds = RS.rb[kKeys];
#ifdef EPSM_CP_REVERSE_ROUNDS            // (A/B build: the rounds of the longest chains first, the window's tail made of light rounds)
#define EPSM_RR(x) (n_rounds - 1 - (x))
#else
#define EPSM_RR(x) (x)
#endif
        LaneId L = RS.lane_of(EPSM_RR(wv), lane);
#ifdef EPSM_CP_PREFETCH
        GeoFetch Xp; AddrFetch Ap;
        fetch_zero(Xp, Ap);
        if (PACKED) own_issue<VARIANT, kList>(Xp, Ap, F, L, WB);           // the window's first round: nothing to hide it under
#endif
#pragma unroll 1
        for (int r = wv; r < n_rounds; r += kWaves) {
            const int q = L.q, c = L.c, k = L.k;
            const uint32_t plan = L.plan;
            const bool ok = plan != 0u, first = k == 1;
            const int64_t i = base + L.loc;                          // lanes without a path touch nothing (every load is guarded)
            const Records<PACKED> R{F, s_ptrs, i};
            const int nv = cp::plan_nv(plan);
            const bool live = ok && q > 0;                           // this lane holds a constraint vertex
            const bool has_next = live && k + 1 <= nv;
            const bool d1 = ok && first && cp::plan_diffuse1(plan);
            const bool act1 = (plan & cp::kPlanActive1) != 0;

            asm volatile("; EPSM_MARK round_begin");
#ifdef EPSM_CP_DRAIN_EACH_ROUND          // (A/B build: what it costs to start every round with an empty queue)
            Q.drain(T);
#endif
            // ---- geometry: own vertex; the two neighbours from the lanes that hold them (or from the words fetched for that)
            cp::Own<float> own;
            own.x = own.e1 = own.e2 = own.n = own.dn1 = own.dn2 = own.light = zero3<float>();
            own.b0 = own.b1 = own.eta = 0.f;
            cp::Nbr<float> prev, next;
            prev.x = prev.e1 = prev.e2 = next.x = next.e1 = next.e2 = zero3<float>();
            V2<float> dk = mk2<float>(0.f, 0.f);
            V3<float> dp = zero3<float>();
            const bool wN = VARIANT == EPSM_VARIANT_MANIFOLD && live && cp::plan_a(plan, k);
            // what only the emission needs (per-field arrays: loaded here; native log: addr_issue, after the recursions)
            float nb0 = 0.f, nb1 = 0.f;                              // barycentrics of vertex k+1
            uint32_t tid_own = kNoIndex, tid_next = kNoIndex, bid = kNoIndex, etri = kNoIndex;
            V3<float> dhf = zero3<float>();
            float eb0 = 0.f, eb1 = 0.f, ew = 0.f;
            U4 sh; sh.x = kNoIndex; sh.y = sh.z = sh.w = 0u;
            if (PACKED) {
                GeoFetch X;                                          // (every field defined: a conditionally loaded, conditionally read
                {                                                    //  struct with undefined fields is kept in scratch memory)
                    const F4v z4 = {0.f, 0.f, 0.f, 0.f};
                    X.o0 = X.o1 = X.o2 = X.o3 = X.o4 = X.o5 = X.p0 = X.p1 = X.p2 = X.n0 = X.n1 = X.n2 = z4;
                    X.o_lz = X.gx = X.gy = 0.f;
                }
#if defined(EPSM_CP_COOP)
                geo_issue_rest<kList>(X, F, L, WB);
                Q.drain(T);                                          // the staging area IS the wave's (now empty) row queue
                geo_stage_coop<VARIANT, kList>(X, F, L, WB, (float *) s_queue[wv], lane);
#elif defined(EPSM_CP_DMA)
                geo_issue_rest<kList>(X, F, L, WB);
#ifdef EPSM_CP_DMA_ALIAS
                Q.drain(T);                                          // the staging area IS the wave's (now empty) row queue
                geo_stage_dma<VARIANT, kList>(X, F, L, WB, (float *) s_queue[wv], lane);
#else
                geo_stage_dma<VARIANT, kList>(X, F, L, WB, s_stage[wv], lane);
#endif
#else
#ifdef EPSM_CP_PREFETCH
#if EPSM_CP_PREFETCH >= 3
                X = Xp;
#elif EPSM_CP_PREFETCH >= 2
                X = Xp;
                if (role_of<kList>(F, L, WB).end_next) { const float *nx = role_of<kList>(F, L, WB).rec + kRecWords; X.n0 = ldq(nx, 0); X.n1 = ldq(nx, 1); X.n2 = ldq(nx, 2); }
#else
                geo_issue_rest<kList>(X, F, L, WB);
                X.o0 = Xp.o0; X.o1 = Xp.o1; X.o2 = Xp.o2; X.o3 = Xp.o3; X.o4 = Xp.o4; X.o5 = Xp.o5;
#endif
#else
                geo_issue<VARIANT, kList>(X, F, L, WB);
#endif
#ifdef EPSM_CP_STASH
                // ONE trip to memory per record: the words only the emission needs (quads 6 and 7: the same line as the geometry,
                // second sector) are requested together with it and parked in LDS until the emission -- 32 bytes per lane in the
                // TAIL of the wave's row queue (items 64..191), which holds fewer than 64 items at this point: the full groups of
                // 64 were drained, the remainder stays at the head.  [q6 | occluder record] [q7]: a lane has either an emitter
                // sample (a `manifold` constraint vertex) or the occluder record (a diffuse first hit), never both.
                Q.drain_full_groups(T);
                {
                    AddrFetch A0;
                    const F4v z4s = {0.f, 0.f, 0.f, 0.f};
                    A0.q6 = A0.q7 = z4s; A0.sh.x = kNoIndex; A0.sh.y = A0.sh.z = A0.sh.w = 0u;
#ifdef EPSM_CP_PREFETCH
                    A0 = Ap;
#else
                    addr_issue<VARIANT, kList>(A0, F, L, WB);
#endif
                    float *st = (float *) s_queue[wv] + 4 * kStashFirstItem + 8 * lane;
                    const bool has_sh = d1 && F.pk_shadow;
                    const F4v shv = {__uint_as_float(A0.sh.x), __uint_as_float(A0.sh.y), __uint_as_float(A0.sh.z), __uint_as_float(A0.sh.w)};
                    *(LdsF4w *) st = has_sh ? shv : A0.q6; *(LdsF4w *) (st + 4) = A0.q7;
                    X.o_lz = A0.q7.x;                                // light.z: word 28
                }
#endif
#endif
                if (live) {
                    const Geo<float> g = geo_from(X.o0, X.o1, X.o2);
                    const Nrm<float> nr = nrm_from(X.o3, X.o4, X.o5, g.b0, g.b1);
                    own.x = g.x; own.e1 = g.e1; own.e2 = g.e2; own.b0 = g.b0; own.b1 = g.b1;
                    own.n = nr.n; own.dn1 = nr.dn1; own.dn2 = nr.dn2;
                    own.eta = X.o3.w; own.light = mk3<float>(X.o5.z, X.o5.w, X.o_lz);
                }
                if (d1 && !live) { own.b0 = X.o2.y; own.b1 = X.o2.z; }
                if (live || d1) tid_own = __float_as_uint(X.o2.w);
                if (c > 1) {                                         // wave-uniform: paths on several lanes exchange their vertices
                    const V3<float> ux = up1(own.x), ue1 = up1(own.e1), ue2 = up1(own.e2);
                    const V3<float> dx = down1(own.x), de1 = down1(own.e1), de2 = down1(own.e2);
                    if (live && !first) { prev.x = ux; prev.e1 = ue1; prev.e2 = ue2; }
                    if (has_next && k < c) { next.x = dx; next.e1 = de1; next.e2 = de2; }
                }
                if (live && first) prev.x = mk3<float>(X.p0.x, X.p0.y, X.p0.z);
                if (has_next && k == c) {
                    const Geo<float> gq = geo_from(X.n0, X.n1, X.n2);
                    next.x = gq.x; next.e1 = gq.e1; next.e2 = gq.e2;
                    nb0 = gq.b0; nb1 = gq.b1; tid_next = __float_as_uint(X.n2.w);
                }
                if (ok && first) {       // epsm.py:250-272 in registers
                    const V3<float> ro = mk3<float>(X.p0.x, X.p0.y, X.p0.z), rd = mk3<float>(X.p0.w, X.p1.x, X.p1.y),
                                    rdx = mk3<float>(X.p1.z, X.p1.w, X.p2.x), rdy = mk3<float>(X.p2.y, X.p2.z, X.p2.w);
                    V3<float> p0 = zero3<float>(), p1 = p0, p2 = p0;
                    const bool need_tri = act1 && (live || d1);      // (dlduv / dldp of a path with no term are never used)
                    if (need_tri) { p0 = mk3<float>(X.o0.x, X.o0.y, X.o0.z); p1 = mk3<float>(X.o0.w, X.o1.x, X.o1.y); p2 = mk3<float>(X.o1.z, X.o1.w, X.o2.x); }
                    const Tangent t = tangent_from(ro, rd, rdx, rdy, X.gx, X.gy, p0, p1, p2, need_tri);
                    dk = mk2<float>(t.db0, t.db1);
                    dp = t.dp;
                    gd_acc = gd_acc + t.gd;
                }
            } else {
                if (live) {
                    own = R.own(k);
                    if (first) prev.x = R.cam();
                    else { const Geo<float> g = R.geo(k - 1); prev.x = g.x; prev.e1 = g.e1; prev.e2 = g.e2; }
                    tid_own = R.tri_id(k);
                    if (wN) R.emit(k, etri, eb0, eb1, ew);
                    if (F.galpha) R.aux(k, bid, dhf);
                }
                if (has_next) {
                    const Geo<float> g = R.geo(k + 1);
                    next.x = g.x; next.e1 = g.e1; next.e2 = g.e2; nb0 = g.b0; nb1 = g.b1;
                    tid_next = R.tri_id(k + 1);
                }
                if (d1) {
                    R.b0b1(1, own.b0, own.b1);
                    tid_own = R.tri_id(1);
                    if (s_ptrs.s[0].shadow) sh = load_u4(s_ptrs.s[0].shadow, i);
                }
                // the tangents of the path (lane k == 1): epsm.py:250-272 in registers, or the caller's arrays
                if (DMODE == kTangentsInKernel) {
                    if (ok && first) {
                        const Tangent t = first_vertex_tangent(F.tin, i, s_ptrs.v[0].p0, s_ptrs.v[0].p1, s_ptrs.v[0].p2, act1);
                        dk = mk2<float>(t.db0, t.db1);
                        dp = t.dp;
                        gd_acc = gd_acc + t.gd;
                    }
                } else {
                    if (ok && (first || DMODE == kTangentsFullRows)) dk = load_d<float, DMODE == kTangentsFullRows>(F.g, i, k, dcols);
                    if (ok && first) dp = load3(F.g.dldp, i);
                }
            }
            // what the second sweep needs of the geometry: positions only (epsm_cp_core.h, Pts); the flat-normal rows need the
            // triangle's edges, the normal rows its interpolated normal (Emitter::vertex)
            cp::Pts<float> pts;
            pts.x = own.x; pts.n = own.n; pts.light = own.light; pts.eta = own.eta; pts.xp = prev.x; pts.xn = next.x;
            const V3<float> keep_e1 = own.e1, keep_e2 = own.e2;
            const float kb0 = own.b0, kb1 = own.b1;

            asm volatile("; EPSM_MARK solve");
            V3<float> Gx = zero3<float>(), gn = Gx, gm = Gx, glight = Gx, gdiff = Gx;
            bool emit_vertex = live;
            AddrFetch A;
            {
                const F4v z4 = {0.f, 0.f, 0.f, 0.f};
                A.q6 = A.q7 = z4;
                A.sh.x = kNoIndex; A.sh.y = A.sh.z = A.sh.w = 0u;
            }
#ifdef EPSM_CPKO_NOSOLVE
            gd_acc.x += own.x.x + own.n.y + own.light.z + own.eta + prev.x.x + prev.e1.y + next.x.z + next.e2.x + dk.x + dp.y;
            if (PACKED) addr_issue<VARIANT, kList>(A, F, L, WB);
            if (false) {
#else
            if (VARIANT == EPSM_VARIANT_MANIFOLD) {
#endif
                cp::MSeeds<float> sd;
                sd.sN = sd.sC = mk2<float>(0.f, 0.f); sd.useN = sd.useC = sd.fC = false;
                if (q > 0) {
                    const bool wC = live && cp::plan_b(plan, k);
                    // pass 1: the 2x2 blocks of the lane's constraint(s)
                    const cp::MBlocks<float> e = cp::manifold_blocks(own, prev, next, wN, has_next);
                    cp::MFwd<float> f = cp::manifold_fwd(e, dk, true, cp::mfwd_zero<float>(), e.Aup, wN, has_next);
#pragma unroll 1
                    for (int s = 2; s <= c; ++s) {                   // forward recursion: lanes with k == s take their step
                        cp::MFwd<float> pf;
                        pf.z = up1(f.z); pf.Sinv = up1(f.Sinv); pf.zN = mk2<float>(0.f, 0.f);
                        const M2<float> pAup = up1(e.Aup);
                        const cp::MFwd<float> g = cp::manifold_fwd(e, dk, false, pf, pAup, wN, has_next);
                        if (k == s) f = g;
                    }
                    cp::MBwd<float> mine; mine.q = mk2<float>(0.f, 0.f); mine.W = 0;
#pragma unroll 1
                    for (int s = c; s >= 1; --s) {                   // backward recursion of the adjoint seeds, on the blocks alone
                        cp::MBwd<float> nb;
                        nb.q = mk2<float>(down1(mine.q.x), down1(mine.q.y)); nb.W = down1(mine.W);
                        if (s == c) { nb.q = mk2<float>(0.f, 0.f); nb.W = 0; }
                        cp::MBwd<float> m2;
                        const cp::MSeeds<float> s2 = cp::manifold_bwd(e, f, nb, wN, wC, has_next, m2);
                        if (k == s) { sd = s2; mine = m2; }
                    }
                }
                // the words only the emission needs, on their way (vector-cache hits) under the second sweep
                __builtin_amdgcn_sched_barrier(0);
#ifdef EPSM_CP_STASH
                if (PACKED) {
                    const float *st = (const float *) s_queue[wv] + 4 * kStashFirstItem + 8 * lane;
                    const F4v a = *(LdsF4 *) st;
                    A.q7 = *(LdsF4 *) (st + 4);
                    if (d1 && F.pk_shadow) { A.sh.x = __float_as_uint(a.x); A.sh.y = __float_as_uint(a.y); A.sh.z = __float_as_uint(a.z); A.sh.w = __float_as_uint(a.w); }
                    else A.q6 = a;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the emission's pushes may overwrite the stash
                }
#else
                if (PACKED) addr_issue<VARIANT, kList>(A, F, L, WB);
#endif
                __builtin_amdgcn_sched_barrier(0);
                if (q > 0) {
                    // pass 2: the constraint(s) swept once more with the final seeds
                    const cp::MOut<float> o = cp::manifold_contract(pts, sd, has_next);
                    const V3<float> from_next = down1(o.GP);         // d/dx_k through constraint k+1
                    Gx = k < c ? o.Gx - from_next : o.Gx;
                    gn = o.gn; gm = o.gm; glight = o.glight; gdiff = o.gdiff;
                }
            } else {
                cp::CFwd<float> f = cp::cfwd_zero<float>();
                bool poisoned = false;
                const int idstar = cp::plan_idstar(plan);
                if (q > 0) {
                    const cp::CBlocks<float> e = cp::caustic_blocks(own, prev, next, first);
                    f = cp::caustic_fwd(e, dk, true, cp::cfwd_zero<float>(), e.Aup);
#pragma unroll 1
                    for (int s = 2; s <= c; ++s) {
                        cp::CFwd<float> pf = cp::cfwd_zero<float>();
                        pf.v = up1(f.v); pf.r = up1(f.r);
                        const M2<float> pAup = up1(e.Aup);
                        const cp::CFwd<float> g = cp::caustic_fwd(e, dk, false, pf, pAup);
                        if (k == s) f = g;
                    }
                    // a non-finite term at id* drops every parameter row of the path (nan_to_num, epsm.py:1076-1079)
                    const unsigned long long bad = __ballot(live && k == idstar && !f.fin);
                    const unsigned long long seg = ((1ull << c) - 1ull) << (lane - (k - 1));
                    poisoned = (bad & seg) != 0ull;
                }
                __builtin_amdgcn_sched_barrier(0);
#ifdef EPSM_CP_STASH
                if (PACKED) {
                    const float *st = (const float *) s_queue[wv] + 4 * kStashFirstItem + 8 * lane;
                    const F4v a = *(LdsF4 *) st;
                    A.q7 = *(LdsF4 *) (st + 4);
                    if (d1 && F.pk_shadow) { A.sh.x = __float_as_uint(a.x); A.sh.y = __float_as_uint(a.y); A.sh.z = __float_as_uint(a.z); A.sh.w = __float_as_uint(a.w); }
                    else A.q6 = a;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the emission's pushes may overwrite the stash
                }
#else
                if (PACKED) addr_issue<VARIANT, kList>(A, F, L, WB);
#endif
                __builtin_amdgcn_sched_barrier(0);
                if (q > 0) {
                    const cp::COut<float> o = cp::caustic_finish(pts, f, first, live && k <= idstar, live && k == idstar, live && cp::plan_b(plan, k));
                    const V3<float> from_next = down1(o.gxp_prev);
                    Gx = k < c ? o.Gx + from_next : o.Gx;
                    gn = o.gn; gm = o.gm; gdiff = o.gdiff;
                    emit_vertex = live && k <= idstar && !poisoned;
                }
            }
            // ---- addressing of the rows this lane will emit: the rows of the scene table its ids name, requested before the
            // next round's touches and the bookkeeping around them
            asm volatile("; EPSM_MARK addressing");
            float fb0 = kb0, fb1 = kb1;                              // barycentrics of the lane's own vertex
            if (PACKED) {
                if (live) {
                    if (F.galpha) dhf = mk3<float>(A.q7.y, A.q7.z, A.q7.w);
                    if (wN) { etri = __float_as_uint(A.q6.x); eb0 = A.q6.y; eb1 = A.q6.z; ew = A.q6.w; }      // (caustic: light_grad == 0)
                }
                if (d1 && F.pk_shadow) sh = A.sh;
                if (c > 1) {                                         // wave-uniform
                    const float db0 = down1(kb0), db1 = down1(kb1);
                    const uint32_t dt = down1(tid_own);
                    if (has_next && k < c) { nb0 = db0; nb1 = db1; tid_next = dt; }
                }
            }
            const U4 t_own = table_row(F.tab, tid_own), t_next = table_row(F.tab, tid_next);
            const U4 er = table_row(F.tab, etri), t_sh = table_row(F.tab, sh.x);
            asm volatile("; EPSM_MARK prefetch");
            // (Until round 4's counter run a "touch" stood here: one word per cache line of the NEXT round's records, so that the
            // lines travelled HBM -> L2 during the emission.  It cost more than it hid: a CU's vector L1 keeps ~64 misses in
            // flight, every 64 bytes a wave reads is one of them for as long as its latency, and a touched line is fetched TWICE
            // through that queue -- by the touch at HBM latency, again by the round that uses it at L2 latency, the L1 having lost
            // it in between.  TCP_PENDING_STALL_CYCLES: 65 % of the kernel; without the touch 2.046 -> 1.970 ms.)
            const LaneId Ln = RS.lane_of(EPSM_RR(r + kWaves), lane);  // (past the last round: no lane has a path)
#ifdef EPSM_CP_PREFETCH
            // (issued here, behind the table rows' loads -- vmcnt counts in order: waiting for those does not wait for these -- and
            // not earlier: before the solve the same 36 registers spill 117)
            if (PACKED) { fetch_zero(Xp, Ap); own_issue<VARIANT, kList>(Xp, Ap, F, Ln, WB); }
#endif
            // ---- emission
            asm volatile("; EPSM_MARK emit");
            if (PACKED) bid = (t_own.w >> 8) - 1u;                    // packed log: alpha slot + 1 in the table row
#ifdef EPSM_CPKO_NOEMIT
            gd_acc.x += Gx.x + gn.y + gm.x + glight.z + gdiff.x + dp.x + (float) (t_own.x + t_next.y + er.z + bid + t_sh.x) + dhf.x + eb0 + eb1 + ew + (emit_vertex ? 1.f : 0.f);
#else
            E.vertex(emit_vertex && q > 0, live, d1, Gx, gn, gm, glight, gdiff, dp, pts.n, keep_e1, keep_e2, fb0, fb1,
                     nb0, nb1, bid, dhf, eb0, eb1, ew, t_own, t_next, er, sh, t_sh);
#endif
            asm volatile("; EPSM_MARK round_end");
            L = Ln;
        }
        Q.drain(T);
        // workgroup-uniform census once per window; a table that fills up in between sends the overflow straight to HBM
        // (not after the workgroup's last window: the final flush follows at once)
        // (flushed after EVERY window -- it holds the next one's rows anyway only when it is far from full -- so that a row's
        // fixed-point sum is bounded by one window's terms: epsm_wave_scatter.h, AccFixed64::kLimit; a launch has more windows
        // than workgroups only beyond 2^31 paths)
        if (wi + 1 < windows_per_block && win + 1 < n_windows) T.flush();
    }
    T.flush(true);
    if (DMODE == kTangentsInKernel && F.grad_o_sum) {      // epsm.py:260-261: d/d ray.o = -sum grad_d, one atomic triple per workgroup
        __shared__ float s_part[kWaves][3];
        float sx = -gd_acc.x, sy = -gd_acc.y, sz = -gd_acc.z;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { sx += __shfl_down(sx, off, 64); sy += __shfl_down(sy, off, 64); sz += __shfl_down(sz, off, 64); }
        if (lane == 0) { s_part[wv][0] = sx; s_part[wv][1] = sy; s_part[wv][2] = sz; }
        __syncthreads();
        if (threadIdx.x < 3) {
            float tot = 0.f;
            for (int w = 0; w < kWaves; ++w) tot += s_part[w][threadIdx.x];
            atomicAdd((my_rep ? my_rep + 6 * F.V + F.B : F.grad_o_sum) + threadIdx.x, tot);
        }
    }
    // ---- small wavefronts in ONE launch (EPSM_OPT_ONE_LAUNCH, off by default): the last workgroup to have flushed into a replica
    //      adds it to the caller's buffers and leaves it zero for the next launch (no workgroup waits for another: the counter says
    //      who is last).  Measured SLOWER than the second, reducing kernel -- 0.144 against 0.086 ms at 524 288 paths: every
    //      workgroup's release fence is an L2 write-back on a part whose eight L2s are not coherent with each other.
    if (my_rep && F.rep_done) {
        __shared__ int s_last;
        const unsigned rix = blockIdx.x % (unsigned) F.replicas;
        __threadfence();                                             // this workgroup's atomics have reached the L2 ...
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned members = ((unsigned) F.rep_blocks - rix + (unsigned) F.replicas - 1u) / (unsigned) F.replicas;
            s_last = atomicAdd(F.rep_done + rix, 1u) + 1u == members;  // ... before it is counted
        }
        __syncthreads();
        if (s_last) {
            __threadfence();
            const int64_t n = 6 * F.V + F.B + 3;
            constexpr int kU = 8;
            for (int64_t e0 = threadIdx.x; e0 < n; e0 += (int64_t) kThreads * kU) {
                float v[kU];
#pragma unroll
                for (int u = 0; u < kU; ++u) {                       // (agent-scope loads: the sums were made by atomics at the L2)
                    const int64_t e = e0 + (int64_t) u * kThreads;
                    v[u] = e < n ? __hip_atomic_load(my_rep + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
                }
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const int64_t e = e0 + (int64_t) u * kThreads;
                    if (e < n && v[u] != 0.f) {
                        float *dst = e < 3 * F.V ? F.gpos + e : e < 6 * F.V ? F.gnrm + (e - 3 * F.V) : e < 6 * F.V + F.B ? (F.galpha ? F.galpha + (e - 6 * F.V) : nullptr)
                                                                                                         : (F.grad_o_sum ? F.grad_o_sum + (e - 6 * F.V - F.B) : nullptr);
                        if (dst && adds_something(v[u])) atomicAdd(dst, v[u]);      // (a non-finite sum adds nothing, as global_add)
                        my_rep[e] = 0.f;
                    }
                }
            }
            if (threadIdx.x == 0) F.rep_done[rix] = 0u;
        }
    }
}

template <int VARIANT, int DMODE, bool PACKED>
hipError_t launch(const FusedArgs &F0, int dcols, hipStream_t s) {
    FusedArgs F = F0;
    // Window size: 2048 paths; a small wavefront (the reference's own backward sizes are 16 384 .. 524 288 paths) is cut so that
    // the ~768 workgroups the chip holds at a time (three per CU) all get ONE window: a wave works through its rounds of a
    // window one after the other, a few microseconds each, and that latency is the run time of a launch with fewer windows
    // than workgroup slots.  (epsm_set_option(EPSM_OPT_SMALL_WAVEFRONT_PATHS) moves the switch: tests.)
    const bool small = F.g.N <= fused_option(EPSM_OPT_SMALL_WAVEFRONT_PATHS);
#ifndef EPSM_CP_SLOTS
#define EPSM_CP_SLOTS 768
#endif
    // `manifold_caustic` walked windows of 1024 paths while a row that found no slot in the table cost three atomic requests (its paths
    // carry three to four constraint vertices with normal and alpha rows each, and 2048 of them overflow the 960-row table: pool slab
    // 3.18 -> 2.80 ms then).  With one request per such row (epsm_wave_scatter.h, drain_queue) and four probes the larger window wins
    // again -- fewer partly filled rounds: pool slab 2.27 -> 2.18 ms.
#ifndef EPSM_CP_WINDOW_CAUSTIC
#define EPSM_CP_WINDOW_CAUSTIC 2048
#endif
    constexpr int kSmall = 1024, kSlots = EPSM_CP_SLOTS;
    constexpr int kLargeRT = VARIANT == EPSM_VARIANT_MANIFOLD ? EPSM_CP_WINDOW : EPSM_CP_WINDOW_CAUSTIC;      // paths per window of a large wavefront
    constexpr int kLarge = kLargeRT > 1024 ? EPSM_CP_WINDOW : 1024;                                         // ... and the instantiation that plans them
    int window = kLargeRT;
    if (small) {
        const int64_t w = ((F.g.N + kSlots - 1) / kSlots + 63) / 64 * 64;       // a multiple of 64 paths
        window = (int) (w < 128 ? 128 : w > kSmall ? kSmall : w);
    }
    const int64_t windows = (F.g.N + window - 1) / window;
    const int64_t blocks = windows < EPSM_CP_BLOCKS ? windows : EPSM_CP_BLOCKS;
    const int64_t per = (windows + blocks - 1) / blocks;
    // small wavefronts: replicas of the gradient buffers (epsm_grad_scatter.hip, DESIGN.md section 5 "Small wavefronts")
    F.rep = nullptr; F.replicas = 1; F.rep_stride = 0; F.rep_done = nullptr; F.rep_blocks = 0;
    if (small) {
        const int64_t stride = (6 * F.V + F.B + 3 + 63) / 64 * 64;        // floats; replicas start on 256-byte boundaries
        int64_t R = blocks / 16;
        if (R > 32) R = 32;
        if (R * stride * 4 > (int64_t) kReplicaBudget) R = (int64_t) kReplicaBudget / (stride * 4);
        if (R >= 4 && fused_option(EPSM_OPT_REPLICAS) != 0) {
            const hipError_t e = fused_workspace(s, (size_t) (R * stride * 4), &F.rep);
            if (e != hipSuccess) return e;
            if (F.rep) {
                F.replicas = (int) R; F.rep_stride = stride;
                F.rep_blocks = (int) blocks;
                F.rep_done = fused_option(EPSM_OPT_ONE_LAUNCH) != 0 ? (uint32_t *) ((char *) F.rep + kReplicaBudget) : nullptr;
            }
        }
    }
    // fixed-point rows hold |sum| < 2^19 at a resolution of 2^-44: fine for terms clamped to +-clip (0.1 in the reference),
    // not for a caller who switched the clamp off
    const bool float_rows = !(F.g.clip <= 1.f);
    // (the small form's instantiation plans at most 1024 paths per window: the planning loops are unrolled over kWindow / 256)
    const bool drop = DMODE != kTangentsInKernel || F.grad_o_sum == nullptr;      // (paths without a term take no lane: kernel, kSortKeys)
#define EPSM_CP_LAUNCH(FLOAT_ROWS_, WINDOW_, DROP_) \
    hipLaunchKernelGGL((epsm_backward_cp_kernel<VARIANT, DMODE, PACKED, FLOAT_ROWS_, WINDOW_, (DROP_) || DMODE != kTangentsInKernel>), \
                       dim3((unsigned) blocks), dim3(kThreads), 0, s, F, dcols, per, window)
    if (float_rows) {
        if (small) { if (drop) EPSM_CP_LAUNCH(true, kSmall, true); else EPSM_CP_LAUNCH(true, kSmall, false); }
        else { if (drop) EPSM_CP_LAUNCH(true, kLarge, true); else EPSM_CP_LAUNCH(true, kLarge, false); }
    } else {
        if (small) { if (drop) EPSM_CP_LAUNCH(false, kSmall, true); else EPSM_CP_LAUNCH(false, kSmall, false); }
        else { if (drop) EPSM_CP_LAUNCH(false, kLarge, true); else EPSM_CP_LAUNCH(false, kLarge, false); }
    }
#undef EPSM_CP_LAUNCH
    hipError_t le = hipGetLastError();
    if (le == hipSuccess && F.rep && !F.rep_done) {
        const int64_t n = 6 * F.V + F.B + 3;
        hipLaunchKernelGGL(reduce_replicas_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, s, F.rep, F.replicas, F.rep_stride,
                           F.V, F.B, F.gpos, F.gnrm, F.galpha, F.grad_o_sum);
        le = hipGetLastError();
    }
    // a launch that did not go through may leave replicas and counters half-way: the workspace is zeroed again before its next use
    if (le != hipSuccess && F.rep) fused_workspace_invalidate(s);
    return le;
}
template <int VARIANT, bool PACKED> hipError_t launch_d(int dmode, const FusedArgs &F, int dcols, hipStream_t s) {
    switch (dmode) {
        case kTangentsTwoColumns: return launch<VARIANT, kTangentsTwoColumns, PACKED>(F, dcols, s);
        case kTangentsFullRows: return launch<VARIANT, kTangentsFullRows, PACKED>(F, dcols, s);
        default: return launch<VARIANT, kTangentsInKernel, PACKED>(F, dcols, s);
    }
}

}  // namespace

namespace epsm {

hipError_t launch_backward_cp(int variant, int dmode, bool packed, const FusedArgs &F, int dcols, hipStream_t s) {
    if (packed) {
        // the native log exists with the tangents computed in the kernel only (epsm_backward_pass_packed)
        return variant == EPSM_VARIANT_MANIFOLD ? launch<EPSM_VARIANT_MANIFOLD, kTangentsInKernel, true>(F, dcols, s)
                                                : launch<EPSM_VARIANT_MANIFOLD_CAUSTIC, kTangentsInKernel, true>(F, dcols, s);
    }
    return variant == EPSM_VARIANT_MANIFOLD ? launch_d<EPSM_VARIANT_MANIFOLD, false>(dmode, F, dcols, s)
                                            : launch_d<EPSM_VARIANT_MANIFOLD_CAUSTIC, false>(dmode, F, dcols, s);
}

}  // namespace epsm
