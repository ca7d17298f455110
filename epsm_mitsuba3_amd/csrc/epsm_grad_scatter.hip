// epsm_grad_scatter.hip -- fused calc_grad + parameter scatter
// (include/epsm.h: epsm_manifold_grad_scatter).
//
// Same per-path arithmetic as epsm_grad.hip (epsm_path_core.h), but the output
// policy feeds every gradient row straight into the workgroup's LDS accumulator
// (epsm_wave_scatter.h) instead of writing calc_grad's dense result lists: the
// 84 B/vertex of (N,3) outputs and their re-read by a scatter pass never touch HBM.
// Persistent workgroups walk contiguous 256-path chunks (neighbouring pixels -> the
// same triangles return -> they stay in the table) and flush to HBM with float
// atomics when the table fills and once at the end.
#include <stdlib.h>
#include <mutex>
#include <type_traits>
#include <stdio.h>
#include <string.h>

#include "epsm_fused.h"
#include "epsm_wave_scatter.h"

using namespace epsm;
using epsm_host::fail;

namespace {

// LDS of a workgroup (two per CU: <= 80 KB each): the accumulator table, four wave queues, and 7.5 KB of window
// flags / permutation / counters / pointer table.  Both sizes matter and neither has to be a power of two (the
// table hashes by multiply-shift).  Measured on config 2 / specular / V = 10^6 (ms; pool and V = 7 829 do not move):
//   2048 rows, 512 items: 4.94 / 18.5 / 5.69      2304, 576: 4.65 / 15.5 / 5.55      2432, 544: 4.71 / 14.4 / 5.38
//   2560, 512: 4.79 / 13.2 / 5.36      3072, 384: 5.00      2048, 384: 5.58      2048, 640: 5.39      1536, 768: 5.51
// (a queue must hold the largest push, 6 rows from 64 lanes = 384 items; the less room beyond that, the more often a
// wave drains a few items with most lanes idle; the table is flushed when a census finds it half full, and a window
// of 1024 paths leaves ~1250 distinct rows on config 2).  Re-measured with the drain inlined: 2304/576 4.5-4.6,
// 2432/544, 2560/512, 2816/448 4.7 (specular 14.5 / 13.3 / 11.7 against 15.9), 2048/640 5.25.
// Short chains (K <= 2 logged vertices: the reference's own backward size, exp/human.py max_depth 3) need <= 168
// VGPRs, so THREE waves per SIMD are possible if three workgroups' LDS fits a CU: a smaller table and queues there.
// Measured at 2^24 paths (config 2 with --vertices K; ms): K = 2: 2.84 -> 2.53, K = 1: 1.43 -> 1.30; from K = 3 on
// the third wave costs spills (K = 3: 168 VGPRs + 240 B of scratch, 3.73 -> 4.21) and the large configuration stays.
// K >= 3, sizes within the 80 KB a workgroup may hold for two per CU (round 2, end): 384-item queues -- the minimum, a push
// is at most 6 rows x 64 lanes -- instead of 576 and the 12 KB they free given to the table: 3072 float rows (2304
// before) / 1920 fixed-point rows (1280): headline slab 3.73 -> 3.67 ms, config 2 4.31 -> 4.19, pool caustic 3.85 -> 3.79.
// (EPSM_AB_*: A/B switches of tools/build_variant.sh.)
#ifndef EPSM_AB_ROWS_FLOAT
#define EPSM_AB_ROWS_FLOAT 3072
#endif
#ifndef EPSM_AB_ROWS_FIXED
#define EPSM_AB_ROWS_FIXED 1920
#endif
#ifndef EPSM_AB_QUEUE
#define EPSM_AB_QUEUE 384
#endif
template <int K> struct Shape {
    static constexpr bool kSmall = K <= 2;
    static constexpr int kRows = kSmall ? 1408 : EPSM_AB_ROWS_FLOAT;        // manifold: table rows, 16 B each (float sums)
    static constexpr int kRowsCaustic = kSmall ? 800 : EPSM_AB_ROWS_FIXED;  // manifold_caustic: rows of 28 B (64-bit fixed-point sums)
    static constexpr int kQueueCap = kSmall ? 384 : EPSM_AB_QUEUE;          // items per wave queue
    static constexpr int kWaves = kSmall ? 3 : 2;                           // waves per SIMD the register budget is set for
};
#ifndef EPSM_FUSED_BLOCKS
#define EPSM_FUSED_BLOCKS 2048
#endif
#ifndef EPSM_SMALL_WAVEFRONT
#define EPSM_SMALL_WAVEFRONT (1 << 20)
#endif
constexpr int64_t kSmallWavefront = EPSM_SMALL_WAVEFRONT;
constexpr int kFusedBlocks = EPSM_FUSED_BLOCKS;             // 512 / 1024 / 8192 measured within 2 %

template <bool FLAGS_IN_LDS, int DMODE, bool PACKED> struct LdsArgs {
    int64_t N;
    const float *cam, *dlduv, *dldp;
    int64_t dlduv_stride;
    V2<float> lane_d;                // kTangentsInKernel: this lane's (d b0, d b1) and d si.p of the current slot
    V3<float> lane_dp;
    const PtrTable *tab;             // LDS
    const uint32_t *win_flags;       // LDS: packed flags of the current window's paths, indexed by path - win_base
    int64_t win_base;
    const float *pk_rays, *pk_verts; // PACKED: the native log
    const uint32_t *pk_flags;
    int pk_K;
    __device__ __forceinline__ const VertexPtrs<float> &vtx(int k) const { return tab->v[k]; }
    __device__ __forceinline__ const float *rec(int k, int64_t i) const { return pk_verts + (i * pk_K + (k - 1)) * kRecWords; }
    // manifold: the window's flags are parked in LDS and a slot starts without a global round trip (-3 %);
    // caustic: re-reading them from the record arrays measured 7 % FASTER than the LDS copy, so it keeps that
    template <int K> __device__ __forceinline__ Flags<K> flags(int64_t i) const {
        if (FLAGS_IN_LDS) return unpack_flags<K>(win_flags[i - win_base]);
        if (PACKED) return unpack_flags<K>(lds_(pk_flags, i));
        return load_flags<float, K>(*this, i);
    }
    __device__ __forceinline__ V3<float> dldp_at(int64_t i) const {
        if (DMODE == kTangentsInKernel) return lane_dp;
        return load3(dldp, i);
    }
    template <bool FULL_D> __device__ __forceinline__ V2<float> d_at(int64_t i, int k, int dcols) const {
        if (DMODE == kTangentsInKernel) return k == 1 ? lane_d : mk2<float>(0.f, 0.f);
        return load_d<float, FULL_D>(*this, i, k, dcols);
    }
    // ---- the record of vertex k of path i (epsm_path_core.h: Raw)
    __device__ __forceinline__ V3<float> cam_at(int64_t i) const {
        if (PACKED) { const F4v q = ldq(pk_rays + 12 * i, 0); return mk3<float>(q.x, q.y, q.z); }
        return load3(cam, i);
    }
    __device__ __forceinline__ Raw<float> raw(int k, int64_t i) const {
        if (!PACKED) return soa_raw<float>(*this, k, i);
        const float *r = rec(k, i);
        const F4v q0 = ldq(r, 0), q1 = ldq(r, 1), q2 = ldq(r, 2), q3 = ldq(r, 3), q4 = ldq(r, 4), q5 = ldq(r, 5);
        Raw<float> o;
        o.g = geo_from(q0, q1, q2, q4.z, q4.w);
        o.nr = nrm_from(q2, q3, q4, q4.z, q4.w);
        o.eta = q5.x;
        o.light = mk3<float>(q5.y, q5.z, q5.w);
        return o;
    }
    __device__ __forceinline__ Geo<float> geo(int k, int64_t i) const {
        if (!PACKED) return load_geo(vtx(k - 1), i);
        const float *r = rec(k, i);
        const F4v q4 = ldq(r, 4);
        return geo_from(ldq(r, 0), ldq(r, 1), ldq(r, 2), q4.z, q4.w);
    }
    __device__ __forceinline__ Nrm<float> nrm(int k, int64_t i, float b0, float b1) const {
        if (!PACKED) return load_nrm(vtx(k - 1), i, b0, b1);
        const float *r = rec(k, i);
        return nrm_from(ldq(r, 2), ldq(r, 3), ldq(r, 4), b0, b1);
    }
    __device__ __forceinline__ float eta(int k, int64_t i) const {
        if (!PACKED) return lds_(vtx(k - 1).eta, i);
        return lds_(rec(k, i), 20);
    }
};

// Output policy: rows go to the LDS table.  The clamp / NaN rule of calc_grad
// (epsm.py:856, 932-944) is applied to each (N,3) component first, exactly as the dense
// path stores it, then the linear map of epsm_scatter_core.h (epsm.py:559-562, 622-627,
// 644-645) follows.  All 64 lanes call every method at the same program point (`ok` is
// false for lanes past the end of the wavefront): rows of the hit triangle are merged
// over runs of equal triangles before they reach LDS.
template <typename Table, bool PACKED, int kQueueCap> struct ScatterOut {
    const FusedArgs &F;
    const PtrTable &P;
    const Table &T;
    WaveQueue<kQueueCap> &Q;
    int64_t i;
    bool ok;

    template <int ROWS>
    __device__ __forceinline__ void push(bool valid, const uint32_t key[ROWS], const V3<float> val[ROWS]) const {
#ifdef EPSM_KO_NOPUSH
        valid = valid && key[0] == 0x12345678u && val[0].x == 1.2345f;
#endif
        Q.reserve(T, ROWS);
        Q.template push_rows<ROWS>(valid, key, val);
    }

    __device__ __forceinline__ const float *rec(int k) const { return F.pk_verts + (i * F.K + (k - 1)) * kRecWords; }
    struct Id { uint32_t v; };
    struct Tri { uint32_t vi[3]; uint32_t mode; };
    struct Emit { uint32_t etri; float eb0, eb1, ew; };
    struct Aux { uint32_t bid; V3<float> dhf; U4 er; float eb0, eb1, ew; };      // er: the emitter triangle's table row

    __device__ __forceinline__ V3<float> fin(V3<float> g) const {
        return mk3<float>(finalize(g.x, F.g.clip), finalize(g.y, F.g.clip), finalize(g.z, F.g.clip));
    }
    __device__ __forceinline__ bool any(bool p) const { return __ballot(p) != 0ull; }
    // triangle id of vertex k: 4 bytes from the path's log, loaded as soon as the path knows which vertices it needs
    __device__ __forceinline__ Id pre_id(int k, bool live) const {
        Id d; d.v = kNoIndex;
#ifdef EPSM_KO_NOADDR
        live = false;
#endif
        // (packed log: reading word 0 of the record along with the id, so that both cache lines of every record the
        // path will need are on their way from the start, measured +0..2 %: the kernel is not waiting for HBM latency)
        if (live && ok) d.v = PACKED ? __float_as_uint(lds_(rec(k), 28)) : lds_(P.s[k - 1].tri, i);
        return d;
    }
    // parameter addressing of vertex k, fetched ahead of the step that needs it: the triangle's row of the scene
    // table (16 B, L2 / MALL resident: 2V rows are a few MB)
    __device__ __forceinline__ Tri pre_tri(int, bool live, Id id) const {
        Tri t; t.vi[0] = t.vi[1] = t.vi[2] = kNoIndex; t.mode = 0;
#ifdef EPSM_KO_NOADDR
        live = false;
#endif
#ifdef EPSM_KO_NOTRI
        live = live && i == -5;
#endif
        if (live && ok) {
            const U4 t4 = table_row(F.tab, id.v);
            t.vi[0] = t4.x; t.vi[1] = t4.y; t.vi[2] = t4.z; t.mode = t4.w;
        }
        return t;
    }
    // emitter-sample record of vertex k: [etri, eb0, eb1, eweight] (k may be a run-time value)
    __device__ __forceinline__ Emit pre_emit(int k, bool live) const {
        Emit e; e.etri = kNoIndex; e.eb0 = e.eb1 = e.ew = 0.f;
#ifdef EPSM_KO_NOADDR
        live = false;
#endif
#ifdef EPSM_KO_NOEMIT
        live = live && i == -5;
#endif
        if (live && ok) {
            if (PACKED) {
                const F4v q = ldq(rec(k), 6);
                e.etri = __float_as_uint(q.x); e.eb0 = q.y; e.eb1 = q.z; e.ew = q.w;
            } else {
                const uint32_t *p = P.s[k - 1].emit;
                if (p) {
                    const U4 e4 = load_u4(p, i);
                    e.etri = e4.x; e.eb0 = bits_to_float(e4.y); e.eb1 = bits_to_float(e4.z); e.ew = bits_to_float(e4.w);
                }
            }
        }
        return e;
    }
    // BSDF record of vertex k and the vertex rows of its emitter triangle (whose id arrived with `e`, a step ago)
    __device__ __forceinline__ Aux pre_aux(int k, bool live, const Emit &e) const {
        Aux a; a.bid = kNoIndex; a.dhf = zero3<float>(); a.eb0 = e.eb0; a.eb1 = e.eb1; a.ew = e.ew;
        a.er.x = a.er.y = a.er.z = kNoIndex; a.er.w = 0u;
#ifdef EPSM_KO_NOADDR
        live = false;
#endif
        if (live && ok) {
            if (PACKED) {
                if (F.galpha) {                      // (the alpha slot comes with the triangle's table row: vertex())
                    const F4v q = ldq(rec(k), 7);
                    a.dhf = mk3<float>(q.y, q.z, q.w);
                }
            } else {
                const ScatterPtrs<float> &s = P.s[k - 1];
#ifndef EPSM_KO_NOAUX
                if (s.aux && F.galpha) {
#else
                if (s.aux && F.galpha && i == -5) {
#endif
                    const U4 a4 = load_u4(s.aux, i);
                    a.bid = a4.x; a.dhf = mk3<float>(bits_to_float(a4.y), bits_to_float(a4.z), bits_to_float(a4.w));
                }
            }
            a.er = table_row(F.tab, e.etri);
        }
        return a;
    }
    __device__ __forceinline__ static bool tri_ok(const Tri &t, int64_t V) {
        return t.vi[0] < (uint64_t) V && t.vi[1] < (uint64_t) V && t.vi[2] < (uint64_t) V;
    }
    __device__ __forceinline__ void vertex(int k, bool has_nm, V3<float> Gx, V3<float> gn, V3<float> gm, V3<float> glight,
                                           const VCtx<float> &c, const Tri &t, const Aux &a) const {
        // epsm.py:559,644: `iteration*5+4 < len(path_grad)` -- the caustic variant never
        // scatters the (always zero) rows of its last vertex
        const float b0 = c.b0, b1 = c.b1, b2 = 1.f - b0 - b1;
        glight = fin(glight);
        const U4 er = a.er;
        V3<float> pos[3] = {fin(Gx * b0), fin(Gx * b1), fin(Gx * b2)};       // si.p_j * path_grad[5it+j]
        V3<float> nrm[3] = {zero3<float>(), zero3<float>(), zero3<float>()};
        gn = fin(gn);
        const bool idx_ok = ok && has_nm && tri_ok(t, F.V);
        const bool pos_v = idx_ok && (t.mode & kModePos);
        bool nrm_v = false;
        if (idx_ok && nz3(gn)) {                                              // si_follow.sh_frame.n * path_grad[5it+3]
            const float sgn = (t.mode & kModeFlip) ? -1.f : 1.f;
            if (t.mode & kModeVertexNormals) {
                if (t.mode & kModeNrm) {
                    // logged normals are post-flip: c.n = sum_j b_j n'_j; sh = normalize(c.n)   (mesh.cpp:784-790, 820-827)
                    const float il = rsqrt_(dot(c.n, c.n));
                    const V3<float> sh = c.n * il;
                    const V3<float> pg = (gn - sh * dot(sh, gn)) * (il * sgn);
                    nrm[0] = pg * b0; nrm[1] = pg * b1; nrm[2] = pg * b2;
                    nrm_v = true;
                }
            } else if (pos_v) {
                // flat: sh = sgn normalize(cross(p1-p0, p2-p0)) with p1-p0 = e2-e1, p2-p0 = -e1   (mesh.cpp:729, 811)
                const V3<float> d0 = c.e2 - c.e1, d1 = -c.e1;
                const V3<float> cr = cross(d0, d1);
                const float il = rsqrt_(dot(cr, cr));
                const V3<float> ch = cr * il;
                const V3<float> cb = (gn - ch * dot(ch, gn)) * (il * sgn);
                const V3<float> d0b = cross(d1, cb), d1b = cross(cb, d0);
                pos[1] = pos[1] + d0b; pos[2] = pos[2] + d1b; pos[0] = pos[0] - (d0b + d1b);
            }
        }
        // group A: the hit triangle's position rows and normal rows (zero rows are dropped at the drain)
        {
            const uint32_t V32 = (uint32_t) F.V;
            const V3<float> z = zero3<float>();
            V3<float> vals[6] = {pos_v ? pos[0] : z, pos_v ? pos[1] : z, pos_v ? pos[2] : z,
                                 nrm_v ? nrm[0] : z, nrm_v ? nrm[1] : z, nrm_v ? nrm[2] : z};
            bool any = (pos_v && (nz3(pos[0]) || nz3(pos[1]) || nz3(pos[2]))) || nrm_v;
            const uint32_t tri[3] = {t.vi[0], t.vi[1], t.vi[2]};
            merge_equal<6, 2>(any, tri, vals, __ballot(nrm_v) != 0ull ? 6 : 3);
            const uint32_t keys[6] = {tri[0], tri[1], tri[2], V32 + tri[0], V32 + tri[1], V32 + tri[2]};
            push<6>(any, keys, vals);
        }
        // group B: bsdf_sample.hf * path_grad[5it+4]  and  si_direct.p * light_grad[it] * sum(Lr_dir)  (epsm.py:622-627, 645)
        {
            gm = fin(gm);
            const uint32_t bid = PACKED ? (t.mode >> 8) - 1u : a.bid;          // packed log: alpha slot + 1 in the table row
            const bool a_ok = ok && has_nm && nz3(gm) && bid < (uint64_t) F.B;
            const bool e_ok = ok && nz3(glight) && er.x < (uint64_t) F.V && er.y < (uint64_t) F.V && er.z < (uint64_t) F.V && (er.w & kModePos);
            const V3<float> gl = e_ok ? glight * a.ew : zero3<float>();
            const uint32_t keys[4] = {e_ok ? er.x : 0u, e_ok ? er.y : 0u, e_ok ? er.z : 0u,
                                      a_ok ? 2u * (uint32_t) F.V + bid : 0u};
            V3<float> vals[4] = {gl * a.eb0, gl * a.eb1, gl * (1.f - a.eb0 - a.eb1),
                                 mk3<float>(a_ok ? dot(gm, a.dhf) : 0.f, 0.f, 0.f)};
            bool e_any = e_ok, a_any = a_ok;
            merge_equal<3, 2>(e_any, keys, vals);                       // area lights are a handful of triangles
            const uint32_t aid[3] = {keys[3], 0u, 0u};
            merge_equal<1, 4>(a_any, aid, vals + 3);                    // a handful of materials
            push<4>(a_any || e_any, keys, vals);
        }
    }
    // si_follow.p * diffuse_grad[it] with detached barycentrics (epsm.py:561-562); vertex it+1
    __device__ __forceinline__ void diffuse(int idx, V3<float> g, float b0, float b1, const Tri &t) const {
        g = fin(g);
        bool pos_v = ok && nz3(g) && idx + 1 <= F.K && tri_ok(t, F.V) && (t.mode & kModePos);
        V3<float> pos[3] = {g * b0, g * b1, g * (1.f - b0 - b1)};
        merge_equal<3, 2>(pos_v, t.vi, pos);
        push<3>(pos_v, t.vi, pos);
    }
    __device__ __forceinline__ void diffuse_first(V3<float> g, Id id) const {
        g = fin(g);
        Tri t = pre_tri(1, nz3(g), id);
        float b0 = 0.f, b1 = 0.f;
        if (ok && nz3(g)) {
            if (PACKED) { const F4v q = ldq(rec(1), 4); b0 = q.z; b1 = q.w; }
            else { b0 = lds_(P.v[0].b0, i); b1 = lds_(P.v[0].b1, i); }
        }
        diffuse(0, g, b0, b1, t);
        // occluder of the first vertex's emitter sample: si_direct.p * diffuse_grad[0] * dis (epsm.py:609-620)
        const uint32_t *shadow = PACKED ? F.pk_shadow : P.s[0].shadow;
        if (shadow) {
            ShadowItems<float> sh = shadow_items<float>(shadow, F.tab, i, ok ? g : zero3<float>(), F.V);
            merge_equal<3, 2>(sh.ok, sh.si, sh.val);
            push<3>(sh.ok, sh.si, sh.val);
        }
    }
    // caustic_path: some lane's term at id* turned out non-finite after rows of its earlier vertices had gone into
    // the accumulator -> the whole wave takes a second turn in which those lanes emit the same rows negated, so the
    // sums end up where the dense route (which zeroes the path's rows, DenseOut::undo_needed) puts them.
    __device__ __forceinline__ bool undo_needed(bool poisoned, int) const { return __ballot(poisoned && ok) != 0ull; }
};

}  // namespace

namespace {

template <int K, int VARIANT, int DMODE, bool PACKED, int kWindow>
// waves-per-SIMD 2: without it hipcc budgets 128 VGPRs from the LDS-derived occupancy and spills 720 B/lane
__global__ __launch_bounds__(256, Shape<K>::kWaves) void epsm_grad_scatter_kernel(FusedArgs F, int dcols, int64_t windows_per_block) {
    constexpr int kQueueCap = Shape<K>::kQueueCap;
    // float rows only where the window's distinct rows need the larger table (epsm_wave_scatter.h, AccFixed64): a 256-path
    // window of a small wavefront fills a few hundred rows, and the integer atomic inserts ~18x faster
#ifdef EPSM_AB_SMALL_FLOAT
    constexpr bool kFloatRows = VARIANT == EPSM_VARIANT_MANIFOLD;
#elif defined(EPSM_AB_ALLFIXED)
    constexpr bool kFloatRows = false;
#else
    constexpr bool kFloatRows = VARIANT == EPSM_VARIANT_MANIFOLD && kWindow == 1024;
#endif
    typedef LdsTable<kFloatRows ? Shape<K>::kRows : Shape<K>::kRowsCaustic,
                     typename std::conditional<kFloatRows, AccFloat, AccFixed64>::type> Table;
    constexpr int kTableSize = Table::kTableSize;
    __shared__ uint32_t s_keys[kTableSize];
    __shared__ typename Table::Val s_vals[kTableSize * 3];
    __shared__ int s_used;
    __shared__ QItem s_queue[4][kQueueCap];
    __shared__ PtrTable s_ptrs;
    float *const my_rep = F.rep ? F.rep + (blockIdx.x % (unsigned) F.replicas) * F.rep_stride : nullptr;
    const Table T{s_keys, s_vals, &s_used, my_rep ? my_rep : F.gpos, my_rep ? my_rep + 3 * F.V : F.gnrm,
                  my_rep ? my_rep + 6 * F.V : F.galpha, (uint32_t) F.V};
    WaveQueue<kQueueCap> Q{s_queue[threadIdx.x >> 6], 0};
    if (!PACKED && threadIdx.x < K) { s_ptrs.v[threadIdx.x] = F.g.v[threadIdx.x]; s_ptrs.s[threadIdx.x] = F.s[threadIdx.x]; }
    __shared__ uint32_t s_flags[VARIANT == EPSM_VARIANT_MANIFOLD ? kWindow : 1];
    constexpr bool FULL_D = DMODE == kTangentsFullRows;
    LdsArgs<VARIANT == EPSM_VARIANT_MANIFOLD, DMODE, PACKED> A{F.g.N, F.g.cam, F.g.dlduv, F.g.dldp, F.g.dlduv_stride,
                                                               mk2<float>(0.f, 0.f), zero3<float>(), &s_ptrs, s_flags, 0,
                                                               F.pk_rays, F.pk_verts, F.pk_flags, F.K};
    V3<float> gd_acc = zero3<float>();           // kTangentsInKernel: sum of grad_d over this lane's paths
    T.clear();                                   // ends with a barrier: the table of pointers is visible too
    // A workgroup takes WINDOWS of 1024 consecutive paths (16 pixels at 64 spp share triangles), a contiguous range
    // of them per workgroup.  Each 256-path sub-chunk of a window is counting-sorted (stable) by the number of
    // vertices its paths are live in and cut into four 64-path slots; in step g the four waves take the four
    // slots of sub-chunk g, rotated so that every wave meets each length class once per window.  A wave then
    // holds paths of (nearly) ONE length -- a step nobody needs is skipped by the whole wave (the flags are
    // independent per path in the worst case: 41 % lane utilisation unsorted) -- the four SIMDs stay balanced
    // without a barrier between steps, and the cache lines of a sub-chunk are touched by the four waves at about
    // the same time (sorting the whole window at once re-fetched every line ~2x: 19 GB instead of 10.8 GB).
    constexpr int kSlots = kWindow / 64, kKeys = K + 1, kSub = kWindow / 256;
    __shared__ uint16_t s_perm[kWindow];
    __shared__ int s_sort;
    __shared__ int s_cnt[kKeys * kSub * 4];            // [key][sub-chunk j][wave w]: histogram, then offsets
    const int64_t n_windows = (F.g.N + kWindow - 1) / kWindow;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll 1
    for (int64_t wi = 0; wi < windows_per_block; ++wi) {
        // a workgroup walks a CONTIGUOUS range of windows: neighbouring pixels keep hitting the triangles whose rows
        // the table already holds, so it fills more slowly and a flushed row carries more (dealt round-robin over the
        // workgroups, as in round 1: +1.5..2.5 %; with the flush threshold at 6/8 instead of 4/8: headline slab
        // 4.01 -> 3.77 ms, config 2 4.46 -> 4.33; the flush atomics are 0.46 ms of the kernel)
        // (also tried: PINNED rows -- emitter vertices and alpha slots, the targets every path adds to, kept in the table
        // across the periodic flushes so that they cost one same-address global atomic per workgroup instead of one per
        // flush: 3.67 -> 3.66 ms, nothing.)
        const int64_t win = (int64_t) blockIdx.x * windows_per_block + wi;
        if (win >= n_windows) break;                   // workgroup-uniform
        const int64_t base = win * kWindow;
        A.win_base = base;
        // -- histogram of the path lengths: thread t holds paths base + j*256 + t
        int key[kSub], rank[kSub];
        {
            // all flag loads of the window first (clamped index, no branch: a branch per sub-chunk would put its
            // consumer behind it and serialise four HBM round trips), then the keys
            Flags<K> fl[kSub];
#pragma unroll
            for (int j = 0; j < kSub; ++j) {
                const int64_t p = base + j * 256 + threadIdx.x;
                const int64_t pc = p < F.g.N ? p : F.g.N - 1;
                fl[j] = PACKED ? unpack_flags<K>(lds_(F.pk_flags, pc)) : load_flags<float, K>(A, pc);
            }
            if (VARIANT == EPSM_VARIANT_MANIFOLD) {
#pragma unroll
                for (int j = 0; j < kSub; ++j) s_flags[j * 256 + threadIdx.x] = pack_flags<K>(fl[j]);   // read back per slot
            }
#pragma unroll
            for (int j = 0; j < kSub; ++j) {
                const int64_t p = base + j * 256 + threadIdx.x;
                const int e = VARIANT == EPSM_VARIANT_MANIFOLD ? manifold_extent<K>(fl[j]) : caustic_extent<K>(fl[j]);
                key[j] = p < F.g.N ? e : 0;
            }
        }
        // (a separate loop: the LDS stores below must not sit between the flag loads of consecutive sub-chunks)
#pragma unroll
        for (int j = 0; j < kSub; ++j) {
#pragma unroll
            for (int q = 0; q < kKeys; ++q) {
                const unsigned long long m = __ballot(key[j] == q);
                if (key[j] == q) rank[j] = __builtin_amdgcn_mbcnt_hi((unsigned) (m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) m, 0u));
                if (lane == 0) s_cnt[(j * kKeys + q) * 4 + wv] = __popcll(m);
            }
        }
        __syncthreads();
        if (wv == 0) {
            constexpr int kEntries = kKeys * kSub * 4;
            // Is sorting worth its gathers?  Steps the 16 natural 64-path groups would run (max length each) against
            // the steps the paths need (sum of lengths / 64): coherent records (real traces) keep their order.
            int gmax = 0, gsum = 0;
            if (lane < kSub * 4) {
                const int j = lane >> 2, w = lane & 3;
#pragma unroll
                for (int q = 0; q < kKeys; ++q) {
                    const int c = s_cnt[(j * kKeys + q) * 4 + w];
                    if (c > 0) gmax = q;
                    gsum += q * c;
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { gmax += __shfl_xor(gmax, off); gsum += __shfl_xor(gsum, off); }
#ifdef EPSM_KO_NOSORT
            if (lane == 0) s_sort = 0;
#else
            if (lane == 0) s_sort = 64 * gmax * 4 > 5 * gsum;      // natural order costs > 1.25x the sorted one
#endif
            // exclusive scan in (j, key, wave) order: each sub-chunk sorted on its own
            int carry = 0;
#pragma unroll
            for (int q0 = 0; q0 < kEntries; q0 += 64) {
                const int q = q0 + lane;
                const int c = q < kEntries ? s_cnt[q] : 0;
                int inc = c;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(inc, off); if (lane >= off) inc += t; }
                if (q < kEntries) s_cnt[q] = carry + inc - c;
                carry += __shfl(inc, 63);
            }
        }
        __syncthreads();
        const bool sorted = s_sort != 0;
#pragma unroll
        for (int j = 0; j < kSub; ++j)
            s_perm[sorted ? s_cnt[(j * kKeys + key[j]) * 4 + wv] + rank[j] : j * 256 + threadIdx.x] = (uint16_t) (j * 256 + threadIdx.x);
        __syncthreads();
        // -- the wave's four slots
#pragma unroll 1
        for (int g = 0; g < kSlots / 4; ++g) {
            // sub-chunk g, quartile rotated: every wave gets each length class once (one-sub-chunk windows: rotated by window)
            const int slot = g * 4 + ((wv + (kSub > 1 ? g : (int) win)) & 3);
            const int64_t i0 = base + s_perm[slot * 64 + lane];
            const bool ok = i0 < F.g.N;
            const int64_t i = ok ? i0 : F.g.N - 1;     // lanes past the end recompute the last path and add nothing
            const ScatterOut<Table, PACKED, kQueueCap> out{F, s_ptrs, T, Q, i, ok};
            // (Tried: the first lines of the NEXT slot's paths -- rays, record 1 -- requested here, a slot ahead of their use:
            // two more live registers, headline slab 3.74 -> 3.90 ms.)
            if (DMODE == kTangentsInKernel) {              // epsm.py:250-272 for this path, in registers
                const Tangent t = PACKED
                    ? first_vertex_tangent_packed(F.tin, i, F.pk_rays + 12 * i, A.rec(1, i), (lds_(F.pk_flags, i) & 4u) != 0)
                    : first_vertex_tangent(F.tin, i, s_ptrs.v[0].p0, s_ptrs.v[0].p1, s_ptrs.v[0].p2,
                                           gl(s_ptrs.v[0].active)[i] != 0);
                A.lane_d = mk2<float>(t.db0, t.db1);
                A.lane_dp = t.dp;
                if (ok) gd_acc = gd_acc + t.gd;
            }
            if (VARIANT == EPSM_VARIANT_MANIFOLD)
                manifold_path<float, K, FULL_D>(A, i, dcols, out);
            else
                caustic_path<float, K, FULL_D>(A, i, dcols, out);
            Q.drain(T);
        }
        // workgroup-uniform census (three barriers) once per window; a table that fills up in between sends
        // the overflow straight to HBM (LdsTable::add)
#if defined(EPSM_KO_NOCENSUS)
        if ((wi & 3) == 3) T.flush();                 // (knock-out: no census, flush every 4th window)
#elif defined(EPSM_KO_FLUSHALWAYS)
        T.flush();                                    // (knock-out: no census, flush every window)
#else
#ifndef EPSM_AB_CROWD
#define EPSM_AB_CROWD 6
#endif
        // (small form: not after the workgroup's last window, the final flush follows at once.  In the large form the two
        // extra live values cost the K = 5 kernel five more spilled registers, 3.74 -> 3.78 ms.  The kernel sits on that
        // edge: 256 VGPRs + 4 spilled; keeping the sum of grad_d per wave in LDS instead of three registers per lane came
        // out of the register allocator with 24 spilled.)
        if ((kWindow == 1024 || (wi + 1 < windows_per_block && win + 1 < n_windows)) && T.crowded(EPSM_AB_CROWD)) T.flush();
#endif
    }
    T.flush();
    if (DMODE == kTangentsInKernel && F.grad_o_sum) {      // epsm.py:260-261: d/d ray.o = -sum grad_d, one atomic triple per workgroup
        __shared__ float s_part[4][3];
        float sx = -gd_acc.x, sy = -gd_acc.y, sz = -gd_acc.z;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { sx += __shfl_down(sx, off, 64); sy += __shfl_down(sy, off, 64); sz += __shfl_down(sz, off, 64); }
        if ((threadIdx.x & 63) == 0) { s_part[threadIdx.x >> 6][0] = sx; s_part[threadIdx.x >> 6][1] = sy; s_part[threadIdx.x >> 6][2] = sz; }
        __syncthreads();
        if (threadIdx.x < 3)
            atomicAdd((my_rep ? my_rep + 6 * F.V + F.B : F.grad_o_sum) + threadIdx.x, s_part[0][threadIdx.x] + s_part[1][threadIdx.x] + s_part[2][threadIdx.x] + s_part[3][threadIdx.x]);
    }
}

}  // namespace

namespace epsm {

// Sums the replicas of a small wavefront into the caller's buffers and leaves the workspace zero for the next launch.
__global__ __launch_bounds__(256) void reduce_replicas_kernel(float *rep, int replicas, int64_t stride, int64_t V, int64_t B,
                                                              float *gpos, float *gnrm, float *galpha, float *go) {
    const int64_t e = (int64_t) blockIdx.x * 256 + threadIdx.x, n = 6 * V + B + 3;
    if (e >= n) return;
    float sum = 0.f;
    for (int r = 0; r < replicas; ++r) { sum += rep[r * stride + e]; rep[r * stride + e] = 0.f; }
    float *dst = e < 3 * V ? gpos + e : e < 6 * V ? gnrm + (e - 3 * V) : e < 6 * V + B ? (galpha ? galpha + (e - 6 * V) : nullptr)
                                                                                      : (go ? go + (e - 6 * V - B) : nullptr);
    if (dst && sum != 0.f) atomicAdd(dst, sum);      // atomic like every other add into the caller's buffers: launches on other streams may share them
}

// Why replicas: the workgroups of a small wavefront all flush at the same moment, and every one of them holds the rows
// every path adds to (the emitter's vertices, the alpha slots, the camera origin).  Global float atomics on ONE 12-byte row
// retire at 12.7 ns each, on 96 rows at 0.86 G rows/s, on rows drawn from >= 15 000 at 18.6 G rows/s
// (tools/micro/global_atomics.hip): 512 workgroups x 96 emitter rows = 57 us of a 190 us kernel (config 5), 2048
// workgroups four times that.  With the rows of workgroup b in replica b % R the same atomics spread over R x as many
// rows; one small kernel sums the replicas afterwards.  The workspace is the library's, one per (device, stream), kept
// zero between launches.
struct Workspace { int dev; hipStream_t stream; float *p; size_t bytes; };
static Workspace g_ws[16];
static int g_ws_n = 0;
static std::mutex g_ws_mutex;
hipError_t fused_workspace(hipStream_t s, size_t bytes, float **out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    Workspace *w = nullptr;
    for (int i = 0; i < g_ws_n; ++i) if (g_ws[i].dev == dev && g_ws[i].stream == s) w = &g_ws[i];
    if (!w) {
        if (g_ws_n == 16) { *out = nullptr; return hipSuccess; }          // no replicas for a 17th stream
        w = &g_ws[g_ws_n++];
        *w = Workspace{dev, s, nullptr, 0};
    }
    if (w->bytes < bytes) {
        if (w->p) { e = hipFree(w->p); w->p = nullptr; w->bytes = 0; if (e != hipSuccess) return e; }
        e = hipMalloc((void **) &w->p, kReplicaBudget);
        if (e != hipSuccess) { w->p = nullptr; return e; }
        w->bytes = kReplicaBudget;
        e = hipMemsetAsync(w->p, 0, kReplicaBudget, s);
        if (e != hipSuccess) return e;
    }
    *out = w->p;
    return hipSuccess;
}

hipError_t fused_release_workspaces() {
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    hipError_t first = hipSuccess;
    int here = 0;
    (void) hipGetDevice(&here);
    for (int i = 0; i < g_ws_n; ++i) {
        if (!g_ws[i].p) continue;
        hipError_t e = hipSetDevice(g_ws[i].dev);
        if (e == hipSuccess) e = hipFree(g_ws[i].p);
        if (e != hipSuccess && first == hipSuccess) first = e;
    }
    g_ws_n = 0;
    (void) hipSetDevice(here);
    return first;
}

}  // namespace epsm

namespace {

template <int K, int VARIANT, int DMODE, bool PACKED = false>
hipError_t launch(const FusedArgs &F0, int dcols, hipStream_t s) {
    FusedArgs F = F0;
    // the unit of work is a 1024-path window; a small wavefront gets one window per workgroup (round 1 rounded the share
    // up to four windows: the 2^19 paths of the reference's own backward size ran on 128 workgroups, half the chip idle)
    // A wave works through its slots of a window one after the other, so a wavefront of few windows is bound by that
    // latency and leaves SIMDs idle: up to 2^20 paths (the reference's own backward sizes: 16 384 .. 524 288) the windows
    // are 256 paths, one slot per wave.
    // (EPSM_SMALL_WAVEFRONT=<paths> in the environment moves the switch: the tests run both forms at every size)
    int64_t small_limit = kSmallWavefront;
    if (const char *e = getenv("EPSM_SMALL_WAVEFRONT")) small_limit = atoll(e);
    const bool small = F.g.N <= small_limit;
    const int64_t windows = small ? (F.g.N + 255) / 256 : (F.g.N + 1023) / 1024;
    const int64_t blocks = windows < kFusedBlocks ? windows : kFusedBlocks;
    const int64_t per = (windows + blocks - 1) / blocks;
    F.rep = nullptr; F.replicas = 1; F.rep_stride = 0;
    if (small) {
        const int64_t stride = (6 * F.V + F.B + 3 + 63) / 64 * 64;        // floats; replicas start on 256-byte boundaries
        int64_t R = blocks / 16;
        if (R > 32) R = 32;
        if (R * stride * 4 > (int64_t) kReplicaBudget) R = (int64_t) kReplicaBudget / (stride * 4);
        const char *off = getenv("EPSM_NO_REPLICAS");
        if (R >= 4 && !(off && off[0] == '1')) {
            const hipError_t e = fused_workspace(s, (size_t) (R * stride * 4), &F.rep);
            if (e != hipSuccess) return e;
            if (F.rep) { F.replicas = (int) R; F.rep_stride = stride; }
        }
    }
    if (small)
        hipLaunchKernelGGL((epsm_grad_scatter_kernel<K, VARIANT, DMODE, PACKED, 256>), dim3((unsigned) blocks), dim3(256), 0, s, F, dcols, per);
    else
        hipLaunchKernelGGL((epsm_grad_scatter_kernel<K, VARIANT, DMODE, PACKED, 1024>), dim3((unsigned) blocks), dim3(256), 0, s, F, dcols, per);
    if (F.rep) {
        const int64_t n = 6 * F.V + F.B + 3;
        hipLaunchKernelGGL(reduce_replicas_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, s, F.rep, F.replicas, F.rep_stride,
                           F.V, F.B, F.gpos, F.gnrm, F.galpha, F.grad_o_sum);
    }
    return hipGetLastError();
}
template <int VARIANT, int DMODE, bool PACKED = false>
hipError_t launch_k(int K, const FusedArgs &F, int dcols, hipStream_t s) {
    switch (K) {
        case 1: return launch<1, VARIANT, DMODE, PACKED>(F, dcols, s);
        case 2: return launch<2, VARIANT, DMODE, PACKED>(F, dcols, s);
        case 3: return launch<3, VARIANT, DMODE, PACKED>(F, dcols, s);
        case 4: return launch<4, VARIANT, DMODE, PACKED>(F, dcols, s);
        default: return launch<5, VARIANT, DMODE, PACKED>(F, dcols, s);
    }
}

}  // namespace

// Which form of the fused kernel runs: the constraint-parallel one (epsm_backward_cp.hip) unless EPSM_BACKWARD_FORM=path
// asks for the one-lane-per-path kernel of this file (kept for A/B measurements; same sums).
static bool use_cp() {
    static const bool cp = [] { const char *e = getenv("EPSM_BACKWARD_FORM"); return !(e && strcmp(e, "path") == 0); }();
    return cp;
}

// Shared by the two entry points: validates the records, fills FusedArgs.  Returns EPSM_OK or fails with `who` in the text.
static int fill_args(FusedArgs &F, const char *who, int variant, int64_t N, int K, const float *cam,
                     const EpsmVertexRecord *verts, const EpsmScatterRecord *sc, const uint32_t *tri_table, int64_t T, float clip,
                     float *grad_pos, float *grad_nrm, float *grad_alpha, int64_t V, int64_t B) {
    char msg[160];
    auto bad = [&](const char *what) { snprintf(msg, sizeof(msg), "%s: %s", who, what); return fail(EPSM_EINVAL, msg); };
    if (variant != EPSM_VARIANT_MANIFOLD && variant != EPSM_VARIANT_MANIFOLD_CAUSTIC) return bad("unknown variant");
    if (K < 1 || K > EPSM_MAX_VERTICES) return bad("K must be in 1..5");
    if (N < 0) return bad("bad N");
    if (!cam || !verts || !sc || !grad_pos || !grad_nrm) return bad("NULL argument");
    if (V < 0 || B < 0 || 2 * V + B >= 0xFFFFFFFFLL) return bad("bad buffer sizes (need 2V+B < 2^32-1)");
    if (T < 0 || (T > 0 && !tri_table) || (((uintptr_t) tri_table) & 15)) return bad("bad triangle table (T rows of 16 B, 16-byte aligned)");
    memset(&F, 0, sizeof(F));
    F.tab = TriTable{tri_table, T};
    F.g.N = N;
    F.g.cam = cam;
    for (int k = 0; k < K; ++k) {
        const EpsmVertexRecord &v = verts[k];
        const EpsmScatterRecord &s = sc[k];
        if (!v.p0 || !v.p1 || !v.p2 || !v.n0 || !v.n1 || !v.n2 || !v.b0 || !v.b1 || !v.eta || !v.light ||
            !v.bsdf || !v.active || !v.active_em || !v.ismesh || !s.tri)
            return bad("NULL pointer in a vertex / scatter record");
        if (((((uintptr_t) s.aux) | ((uintptr_t) s.emit) | ((uintptr_t) s.shadow)) & 15) || (((uintptr_t) s.tri) & 3))
            return bad("aux/emit/shadow must be 16-byte aligned (tri: 4)");
        VertexPtrs<float> &o = F.g.v[k];
        o.p0 = (const float *) v.p0; o.p1 = (const float *) v.p1; o.p2 = (const float *) v.p2;
        o.n0 = (const float *) v.n0; o.n1 = (const float *) v.n1; o.n2 = (const float *) v.n2;
        o.b0 = (const float *) v.b0; o.b1 = (const float *) v.b1; o.eta = (const float *) v.eta;
        o.light = (const float *) v.light;
        o.bsdf = v.bsdf; o.active = v.active; o.active_em = v.active_em; o.ismesh = v.ismesh;
        ScatterPtrs<float> &t = F.s[k];
        t.tri = s.tri; t.aux = s.aux; t.emit = s.emit;
        t.shadow = k == 0 ? s.shadow : nullptr;       // epsm.py:610: `iteration == 0`
    }
    F.g.clip = (clip > 0.0f && clip <= 3.402823466e+38f) ? clip : 3.402823466e+38f;
    F.gpos = grad_pos; F.gnrm = grad_nrm; F.galpha = grad_alpha;
    F.V = V; F.B = grad_alpha ? B : 0;
    F.P = epsm_num_param_grads(variant, K);
    F.K = K;
    return EPSM_OK;
}

extern "C" int epsm_release_workspace(void) {
    epsm_host::err_buf()[0] = 0;
    const hipError_t e = fused_release_workspaces();
    if (e != hipSuccess) return epsm_host::hip_fail("epsm_release_workspace", e);
    return EPSM_OK;
}

extern "C" int epsm_manifold_grad_scatter(int variant, int64_t N, int K,
                                          const float *cam, const EpsmVertexRecord *verts,
                                          const EpsmScatterRecord *sc, const uint32_t *tri_table, int64_t T,
                                          const float *dlduv, int64_t dlduv_stride, int dlduv_cols,
                                          const float *dldp, float clip,
                                          float *grad_pos, float *grad_nrm, float *grad_alpha,
                                          int64_t V, int64_t B, void *stream) {
    epsm_host::err_buf()[0] = 0;
    if (N == 0 && K >= 1 && K <= EPSM_MAX_VERTICES &&
        (variant == EPSM_VARIANT_MANIFOLD || variant == EPSM_VARIANT_MANIFOLD_CAUSTIC)) return EPSM_OK;
    FusedArgs F;
    const int rc = fill_args(F, "epsm_manifold_grad_scatter", variant, N, K, cam, verts, sc, tri_table, T, clip, grad_pos, grad_nrm, grad_alpha, V, B);
    if (rc != EPSM_OK) return rc;
    if (!dlduv || !dldp) return fail(EPSM_EINVAL, "epsm_manifold_grad_scatter: NULL argument");
    if (dlduv_cols < 0 || dlduv_stride < (dlduv_cols < 2 * K ? dlduv_cols : 2 * K))
        return fail(EPSM_EINVAL, "epsm_manifold_grad_scatter: dlduv_stride smaller than the columns to read");
    F.g.dlduv = dlduv;
    F.g.dlduv_stride = dlduv_stride;
    F.g.dldp = dldp;
    int dcols = dlduv_cols > 2 * K ? 2 * K : dlduv_cols;
    const bool full_d = dcols > 2;
    hipStream_t s = (hipStream_t) stream;
    hipError_t e;
    if (use_cp())
        e = launch_backward_cp(variant, full_d ? kTangentsFullRows : kTangentsTwoColumns, false, F, dcols, s);
    else if (variant == EPSM_VARIANT_MANIFOLD)
        e = full_d ? launch_k<EPSM_VARIANT_MANIFOLD, kTangentsFullRows>(K, F, dcols, s)
                   : launch_k<EPSM_VARIANT_MANIFOLD, kTangentsTwoColumns>(K, F, dcols, s);
    else
        e = full_d ? launch_k<EPSM_VARIANT_MANIFOLD_CAUSTIC, kTangentsFullRows>(K, F, dcols, s)
                   : launch_k<EPSM_VARIANT_MANIFOLD_CAUSTIC, kTangentsTwoColumns>(K, F, dcols, s);
    if (e != hipSuccess) return epsm_host::hip_fail("epsm_manifold_grad_scatter", e);
    return EPSM_OK;
}

extern "C" int epsm_backward_pass(int variant, int64_t N, int K, int64_t path_offset, int spp, int res,
                                  const float *ray_o, const float *ray_d, const float *ray_dx, const float *ray_dy,
                                  const float *grad_img, int img_width, int img_channels,
                                  const EpsmVertexRecord *verts, const EpsmScatterRecord *sc,
                                  const uint32_t *tri_table, int64_t T, float clip,
                                  float *grad_pos, float *grad_nrm, float *grad_alpha, float *grad_o_sum,
                                  int64_t V, int64_t B, void *stream) {
    epsm_host::err_buf()[0] = 0;
    if (N == 0 && K >= 1 && K <= EPSM_MAX_VERTICES &&
        (variant == EPSM_VARIANT_MANIFOLD || variant == EPSM_VARIANT_MANIFOLD_CAUSTIC)) return EPSM_OK;
    FusedArgs F;
    const int rc = fill_args(F, "epsm_backward_pass", variant, N, K, ray_o, verts, sc, tri_table, T, clip, grad_pos, grad_nrm, grad_alpha, V, B);
    if (rc != EPSM_OK) return rc;
    if (!ray_d || !ray_dx || !ray_dy || !grad_img) return fail(EPSM_EINVAL, "epsm_backward_pass: NULL argument");
    if (spp < 1 || res < 1 || img_width < res || img_channels < 5)
        return fail(EPSM_EINVAL, "epsm_backward_pass: need spp>=1, res>=1, img_width>=res, img_channels>=5");
    if (path_offset < 0 || (int64_t) res * res * spp < path_offset + N)
        return fail(EPSM_EINVAL, "epsm_backward_pass: path_offset + N exceeds res*res*spp");
    F.tin = TangentIn{path_offset, spp, res, img_width, img_channels, ray_o, ray_d, ray_dx, ray_dy, grad_img};
    F.grad_o_sum = grad_o_sum;
    hipStream_t s = (hipStream_t) stream;
    const hipError_t e = use_cp() ? launch_backward_cp(variant, kTangentsInKernel, false, F, 2, s)
                       : variant == EPSM_VARIANT_MANIFOLD ? launch_k<EPSM_VARIANT_MANIFOLD, kTangentsInKernel>(K, F, 2, s)
                                                          : launch_k<EPSM_VARIANT_MANIFOLD_CAUSTIC, kTangentsInKernel>(K, F, 2, s);
    if (e != hipSuccess) return epsm_host::hip_fail("epsm_backward_pass", e);
    return EPSM_OK;
}


// The same pass on the native log (include/epsm.h: EpsmPackedLog): one 128-byte record per (path, vertex), the
// rays and the flag word per path.
extern "C" int epsm_backward_pass_packed(int variant, int64_t N, int K, int64_t path_offset, int spp, int res,
                                         const EpsmPackedLog *log, const float *grad_img, int img_width, int img_channels,
                                         const uint32_t *tri_table, int64_t T, float clip,
                                         float *grad_pos, float *grad_nrm, float *grad_alpha, float *grad_o_sum,
                                         int64_t V, int64_t B, void *stream) {
    epsm_host::err_buf()[0] = 0;
    if (variant != EPSM_VARIANT_MANIFOLD && variant != EPSM_VARIANT_MANIFOLD_CAUSTIC) return fail(EPSM_EINVAL, "epsm_backward_pass_packed: unknown variant");
    if (K < 1 || K > EPSM_MAX_VERTICES) return fail(EPSM_EINVAL, "epsm_backward_pass_packed: K must be in 1..5");
    if (N == 0) return EPSM_OK;
    if (N < 0 || !log || !log->rays || !log->flags || !log->verts || !grad_img || !grad_pos || !grad_nrm)
        return fail(EPSM_EINVAL, "epsm_backward_pass_packed: NULL argument / bad N");
    if ((((uintptr_t) log->rays) | ((uintptr_t) log->verts) | ((uintptr_t) log->shadow) | ((uintptr_t) tri_table)) & 15)
        return fail(EPSM_EINVAL, "epsm_backward_pass_packed: rays / verts / shadow / tri_table must be 16-byte aligned");
    if (V < 0 || B < 0 || 2 * V + B >= 0xFFFFFFFFLL || T < 0 || (T > 0 && !tri_table))
        return fail(EPSM_EINVAL, "epsm_backward_pass_packed: bad buffer sizes");
    if (spp < 1 || res < 1 || img_width < res || img_channels < 5)
        return fail(EPSM_EINVAL, "epsm_backward_pass_packed: need spp>=1, res>=1, img_width>=res, img_channels>=5");
    if (path_offset < 0 || (int64_t) res * res * spp < path_offset + N)
        return fail(EPSM_EINVAL, "epsm_backward_pass_packed: path_offset + N exceeds res*res*spp");
    FusedArgs F;
    memset(&F, 0, sizeof(F));
    F.tab = TriTable{tri_table, T};
    F.g.N = N;
    F.g.clip = (clip > 0.0f && clip <= 3.402823466e+38f) ? clip : 3.402823466e+38f;
    F.gpos = grad_pos; F.gnrm = grad_nrm; F.galpha = grad_alpha;
    F.V = V; F.B = grad_alpha ? B : 0;
    F.P = epsm_num_param_grads(variant, K);
    F.K = K;
    F.pk_rays = log->rays; F.pk_flags = log->flags; F.pk_verts = (const float *) log->verts; F.pk_shadow = log->shadow;
    F.tin = TangentIn{path_offset, spp, res, img_width, img_channels, nullptr, nullptr, nullptr, nullptr, grad_img};
    F.grad_o_sum = grad_o_sum;
    hipStream_t s = (hipStream_t) stream;
    const hipError_t e = use_cp() ? launch_backward_cp(variant, kTangentsInKernel, true, F, 2, s)
                       : variant == EPSM_VARIANT_MANIFOLD ? launch_k<EPSM_VARIANT_MANIFOLD, kTangentsInKernel, true>(K, F, 2, s)
                                                          : launch_k<EPSM_VARIANT_MANIFOLD_CAUSTIC, kTangentsInKernel, true>(K, F, 2, s);
    if (e != hipSuccess) return epsm_host::hip_fail("epsm_backward_pass_packed", e);
    return EPSM_OK;
}
