// epsm_trace.hip -- kernels + C ABI of the wavefront tracer (include/epsm_trace.h).
#include <stdio.h>
#include <string.h>

#include "epsm_common.h"
#include "epsm_trace_core.h"
#include "epsm_trace_wavefront.h"
#include "epsm_trace_quad.h"
#include "epsm_trace_packet.h"
#include "epsm_wave_scatter.h"           // wave_total_lane63 (DPP sums): the first-hit stage

using namespace epsm;
using epsm_host::fail;

namespace {

// four waves per SIMD: 128 registers and 56 B/lane of scratch instead of 151 registers and three waves:
// -11 % / -17 % at 128 k / 512 k triangles (five waves, 96 registers + 184 B of scratch: +5 % / -2 %)
__global__ __launch_bounds__(128, 4) void epsm_trace_kernel(TraceArgs A) {
    // traversal stacks: one LDS column of 32 entries per path (16 KB: four waves per SIMD); the four-wide tree may push
    // 3 x 16 references: the rare entries beyond 32 go to a private array (scratch)
    constexpr int kLds = 32;
    __shared__ uint32_t s_stack[kLds * 128];
    uint32_t deep[kBvhStack - kLds];
    const int64_t i = (int64_t) blockIdx.x * 128 + threadIdx.x;
    BvhStack st{s_stack + threadIdx.x, 128};
    st.cap = kLds; st.ovf = deep; st.ovf_stride = 1;
#ifdef EPSM_MEGA_NO_PACKET
    if (i >= A.N) return;
    trace_one_path(A, i, st);
#else
    // trace_one_path with the PRIMARY rays walked by the wave (epsm_trace_packet.h: the lanes of a wave are the samples of one
    // pixel or of a few neighbours); the packet's LDS column is entry 0 of the wave's own per-lane stacks, not yet in use
    const bool has = i < A.N;
    if (__ballot(has) == 0ull) return;
    PathState s = path_begin(A, has ? i : A.N - 1, has);
    const TriHit th0 = packet_intersect<false>(A.S, s.ray, has, s_stack + (threadIdx.x & ~63));
    if (!has) return;
    InlineVis vis{st};
    const int max_depth = path_max_depth(A);
    for (int iteration = 0; iteration < max_depth; ++iteration) {
        TriHit th; th.hit = false; th.tri = 0; th.t = kInf; th.u = th.v = 0.f;
        if (iteration == 0) th = th0;
        else if (s.active) th = intersect<false>(A.S, s.ray, st);
        path_bounce(A, i, iteration, s, th, vis);
    }
    path_end(A, i, s);
#endif
}

// ---- the wavefront form (epsm_trace_wavefront.h): queues of live paths, three small kernels per bounce ----
constexpr int kWfThreads = 128;                         // traversal kernels
constexpr int kWfMaxBlocks = 16384;
constexpr int kWfMaxChunkBlocks = 8192;                 // compaction: workgroups of 256, each takes chunks in turn

// Internal flag (never a caller's): under EPSM_TRACE_FUSE_FIRST_HIT the primary rays' stage retires most paths itself, and the few
// it leaves are COMPACTED into a queue before bounce 0's shade stage (a scan + compaction pass numbered bounce -1) -- slot q of bounce 0
// is then path queue[0][q] of counters[0], not path q of N.
constexpr uint32_t kWfPreCompact = 0x80000000u;
__device__ __forceinline__ bool wf_identity(const TraceArgs &A, int b) { return b < 0 || (b == 0 && !(A.flags & kWfPreCompact)); }   // slot == path
__device__ __forceinline__ int64_t wf_count(const TraceArgs &A, const WfState &W, int b) {
    return wf_identity(A, b) ? A.N : (int64_t) W.counters[b];
}
// Traversal kernels: 8 KB of LDS stacks per workgroup, 47-50 registers: 8 waves per SIMD.
// (Tried and dropped: persistent waves whose idle lanes fetch new rays between two traversal rounds -- from a shared
// queue head the ~10^5 same-address atomics serialise, from a static share per wave the primary rays lose their
// coherence: 667 -> 1226 us for bounce 0, -5..10 % for the later bounces, +1.1 ms per 4.2 M paths in all.
// And: ONE loop whose every turn runs one kind of step -- a node step or a triangle test -- chosen by majority among
// the unfinished lanes (the while-while loops run for the slowest lane: 27 % of the VALU lanes busy).  Same hits;
// 5.6-5.7 ms against 5.3-5.4 ms at 128 k triangles, 8.7 against 8.9 ms at 512 k: the step-wise traversal state
// (leaf cursor kept across turns) costs the while-while form 5-7 % where both share it, so it was not kept.)
// FIRST: bounce 0, whose rays are the primary rays (derived from the path index, nothing read); its own instantiation so
// that the later bounces keep their 47 registers.
template <bool FIRST>
__global__ __launch_bounds__(kWfThreads) void epsm_wf_extend_kernel(TraceArgs A, WfState W, int b) {
    __shared__ uint32_t s_stack[kWfStackLds * kWfThreads];
    const int64_t count = wf_count(A, W, b);
    if (!FIRST && wf_in_tail(A, b, count)) return;
    for (int64_t q = (int64_t) blockIdx.x * kWfThreads + threadIdx.x; q < count; q += (int64_t) gridDim.x * kWfThreads)
        wf_extend(A, W, FIRST ? q : (int64_t) W.queue[b & 1][q], s_stack + threadIdx.x, kWfThreads, FIRST ? 0 : 1);
}
// The closest-hit stage with the WAVE as the unit (epsm_trace_packet.h).  FIRST: the primary rays -- a wave is the samples of one
// pixel or of a few neighbours.
// EPSM_TRACE_FUSE_FIRST_HIT: what the lanes of a wave -- at bounce 0 the samples of one pixel or of a few neighbouring ones -- give the
// backward pass: every path's grad_d into the wave's share of -sum grad_d, and the first-vertex rows of the paths retired at their
// first hit.  Lanes on the same triangle are summed first (butterfly over the wave, in up to four turns of "the first lane still
// owing and everybody on its triangle"), one lane adds the sum; what is left after four turns adds for itself.
#ifdef EPSM_FH_SHUFFLE_SUM                // (A/B build: the butterfly through the LDS crossbar)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
    return v;
}
#else
// sum over the wave by six DPP adds (epsm_wave_scatter.h: the total lands in lane 63) + one readlane: no trip through the LDS crossbar
__device__ __forceinline__ float wave_sum(float v) { return lane63(wave_total_lane63(v)); }
#endif
__device__ __forceinline__ void first_hit_add(float *p, float v) { if (v != 0.f && fabsf(v) < __builtin_inff()) atomicAdd(p, v); }
__device__ __forceinline__ void first_hit_scatter(const TraceArgs &A, const WfState &W, const WfFirstHit &fh, unsigned wave_index) {
    const int lane = threadIdx.x & 63;
    if (A.fh.grad_o_sum) {
        // epsm.py:258-261: d / d ray.o = -sum grad_d.  One triple per WAVE on the three words themselves was 2 ms per 2^24 paths
        // (262 144 same-address float atomics retire one at a time); the waves add to one of 256 slots of the workspace, which
        // epsm_wf_first_hit_finish_kernel sums into the caller's three words.
        const float sx = wave_sum(fh.gd.x), sy = wave_sum(fh.gd.y), sz = wave_sum(fh.gd.z);
        if (lane < 3) first_hit_add(W.fh_partial + 4 * (wave_index % kWfFirstHitSlots) + lane,
                                    lane == 0 ? sx : lane == 1 ? sy : sz);
    }
    bool owing = fh.rows.on;
#pragma unroll 1
    for (int turn = 0; turn < 4; ++turn) {
        const unsigned long long m = __ballot(owing);
        if (m == 0ull) return;
        const int leader = __ffsll((long long) m) - 1;
        const uint32_t k0 = (uint32_t) __shfl((int) fh.rows.key[0], leader), k1 = (uint32_t) __shfl((int) fh.rows.key[1], leader),
                       k2 = (uint32_t) __shfl((int) fh.rows.key[2], leader);
        const bool mine = owing && fh.rows.key[0] == k0 && fh.rows.key[1] == k1 && fh.rows.key[2] == k2;
        float v[9];
#pragma unroll
        for (int j = 0; j < 3; ++j) { v[3 * j] = mine ? fh.rows.val[j].x : 0.f; v[3 * j + 1] = mine ? fh.rows.val[j].y : 0.f; v[3 * j + 2] = mine ? fh.rows.val[j].z : 0.f; }
#pragma unroll
        for (int c = 0; c < 9; ++c) v[c] = wave_sum(v[c]);
        // lanes 0..8 add one component each: x, y, z of a row side by side -> one atomic request per row
        if (lane < 9) {
            const int j = lane / 3, c = lane - 3 * j;
            const uint32_t row = j == 0 ? k0 : j == 1 ? k1 : k2;
            float val = v[0];
#pragma unroll
            for (int e = 1; e < 9; ++e) val = lane == e ? v[e] : val;
            first_hit_add(A.fh.grad_pos + 3 * (int64_t) row + c, val);
        }
        if (mine) owing = false;
    }
    if (owing) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float *p = A.fh.grad_pos + 3 * (int64_t) fh.rows.key[j];
            first_hit_add(p, fh.rows.val[j].x); first_hit_add(p + 1, fh.rows.val[j].y); first_hit_add(p + 2, fh.rows.val[j].z);
        }
    }
}
template <bool FIRST>
__global__ __launch_bounds__(kWfThreads) void epsm_wf_extend_packet_kernel(TraceArgs A, WfState W, int b) {
    __shared__ uint32_t s_stack[kPacketStack * (kWfThreads / 64)];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t count = FIRST ? A.N : wf_count(A, W, b);     // (the primary rays: every path, whatever is compacted behind this stage)
    if (!FIRST && wf_in_tail(A, b, count)) return;
    for (int64_t q0 = (int64_t) blockIdx.x * kWfThreads + wv * 64; q0 < count; q0 += (int64_t) gridDim.x * kWfThreads) {     // wave-uniform
        const int64_t q = q0 + lane;
        const bool has = q < count;
        const int64_t i = FIRST ? (has ? q : q0) : (int64_t) W.queue[b & 1][has ? q : q0];
        Ray r;
        // EPSM_TRACE_FUSE_FIRST_HIT: the primary rays' stage has ray and hit in registers -- it retires the paths the rule retires at
        // their first vertex itself (first_hit_retires / first_hit_rows: the backward pass of a path without a chain, every path's
        // share of d / d ray.o) and marks their slots; the shade stage passes them by: of 2^24 paths of the clutter scene it shades 2 M.
        const bool fuse = FIRST && (A.flags & EPSM_TRACE_FUSE_FIRST_HIT);      // (kernel-uniform)
        WfFirstHit fh;
        fh.rows.on = false; fh.rows.key[0] = fh.rows.key[1] = fh.rows.key[2] = kNoIndex;
        fh.rows.val[0] = fh.rows.val[1] = fh.rows.val[2] = fh.gd = zero3<float>();
        if (FIRST) {
            PrimaryRay pr;
            r = path_begin(A, i, false, &pr).ray;
            if (fuse && has) {
                float gx, gy;
                first_hit_pixel_grad(A, i, gx, gy);
                fh.gd = (pr.dx - pr.ray.d) * gx + (pr.dy - pr.ray.d) * gy;     // epsm.py:255, as tangent_from forms it
            }
        } else {
            const W4 o = W.ray_o[i], d = W.ray_d[i];
            r.o = xyz(o); r.maxt = u2f(o.w); r.d = xyz(d);
        }
        const TriHit th = packet_intersect<false>(A.S, r, has, s_stack + wv * kPacketStack);
        bool done = false;
        if (fuse) {
            if (has) {
                uint32_t w;
                SurfHit lite;
                if (first_hit_retires(A, th, w, lite)) {
                    fh.rows = first_hit_rows(A, w, lite, r, fh.gd);
                    A.rec[0].pflags[i] = 0u;                             // no vertex: the backward kernel gives this path no lane
                    done = true;
                }
                W.flags[q] = done ? (uint8_t) 0 : kWfAlive;              // (the pass numbered bounce -1 compacts the survivors)
            }
            first_hit_scatter(A, W, fh, (unsigned) (q0 >> 6));           // (all lanes of the wave: the sums run over it)
            const unsigned long long ma = __ballot(has && !done);
            if (lane == 0 && ma != 0ull) {                               // the survivors of this quarter of a chunk, for the scan
                const int64_t chunk = q0 / kWfChunk;
                atomicAdd(&W.chunk_counts[chunk], (uint32_t) __popcll(ma));
                atomicAdd(&W.group_counts[chunk / kWfGroup], (uint32_t) __popcll(ma));
            }
        }
        if (has && !done) {
            W4 h; h.x = th.hit ? th.tri : kNoIndex; h.y = f2u(th.t); h.z = f2u(th.u); h.w = f2u(th.v);
            W.hit[i] = h;
        }
    }
}
// The visibility rays the same way (A/B build).
__global__ __launch_bounds__(kWfThreads) void epsm_wf_shadow_packet_kernel(TraceArgs A, WfState W, int b) {
    __shared__ uint32_t s_stack[kPacketStack * (kWfThreads / 64)];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t count = (int64_t) W.counters[8 + b];
    for (int64_t q0 = (int64_t) blockIdx.x * kWfThreads + wv * 64; q0 < count; q0 += (int64_t) gridDim.x * kWfThreads) {     // wave-uniform
        const int64_t q = q0 + lane;
        const bool has = q < count;
        const int64_t i = (int64_t) W.shadow_queue[has ? q : q0];
        const W4 o = W.sh_o[i], d = W.sh_d[i];
        Ray sr; sr.o = xyz(o); sr.maxt = u2f(o.w); sr.d = xyz(d);
        const TriHit th = packet_intersect<true>(A.S, sr, has, s_stack + wv * kPacketStack);
        const bool owed = has && wf_shadow_resolve(A, W, i, b, th.hit);
        if (__ballot(owed) != 0ull) {                                    // (integrators with max_depth <= 3: the occluder record)
            F3 sip = zero3<float>(), esp = sip;
            Ray r2; r2.o = r2.d = sip; r2.maxt = 0.f;
            if (owed) r2 = wf_occluder_ray(A, W, i, sip, esp);
            const TriHit oh = packet_intersect<false>(A.S, r2, owed, s_stack + wv * kPacketStack);
            if (owed) write_occluder(A.S, A.rec[0].shadow + 4 * i, r2, oh, sip, esp);
        }
    }
}
// The rest of the loop of the paths alive into bounce b, once they are few (wf_tail, epsm_trace_wavefront.h).
__global__ __launch_bounds__(kWfThreads) void epsm_wf_tail_kernel(TraceArgs A, WfState W, int b) {
    __shared__ uint32_t s_stack[kWfStackLds * kWfThreads];
    const int64_t count = wf_count(A, W, b);
    if (!wf_in_tail(A, b, count)) return;
    for (int64_t q = (int64_t) blockIdx.x * kWfThreads + threadIdx.x; q < count; q += (int64_t) gridDim.x * kWfThreads)
        wf_tail(A, W, (int64_t) W.queue[b & 1][q], b, s_stack + threadIdx.x, kWfThreads);
}
__global__ __launch_bounds__(kWfFirstHitSlots) void epsm_wf_first_hit_finish_kernel(TraceArgs A, WfState W) {
    __shared__ float s_sum[kWfFirstHitSlots / 64][3];
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = wave_sum(W.fh_partial[4 * threadIdx.x + c]);
    if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6][0] = v[0]; s_sum[threadIdx.x >> 6][1] = v[1]; s_sum[threadIdx.x >> 6][2] = v[2]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        float t = 0.f;
        for (int w = 0; w < kWfFirstHitSlots / 64; ++w) t += s_sum[w][threadIdx.x];
        first_hit_add(A.fh.grad_o_sum + threadIdx.x, -t);
    }
}
// (160 registers, 3 waves per SIMD; capped at 128 for 4 waves it spills 96 B/lane and is no faster.)
// One 256-slot chunk of the queue per workgroup.  Appending the survivors to the next queue with one atomic per
// wave on the queue's counter made this kernel 1.23 ms at 4.2 M paths whatever it computed (knock-outs: the log
// writes cost nothing, each of the two queues 0.6 ms: 65 536 same-address atomics, ~9 ns apiece at the memory
// side); it leaves a flag per slot and two counts per chunk instead.
__global__ __launch_bounds__(kWfChunk) void epsm_wf_shade_kernel(TraceArgs A, WfState W, int b) {
    const int64_t count = wf_count(A, W, b);
    if ((int64_t) blockIdx.x * kWfChunk >= count || wf_in_tail(A, b, count)) return;           // workgroup-uniform
    // (A capped grid whose workgroups take chunks in turn, as the compaction does: the launch of a bounce nobody reaches 15 -> 5 us, but
    // a stage whose every chunk has work 1.08 -> 1.19 ms at 2^24 paths -- the hardware's dispatch order balances better.  So the
    // host caps the grid only where the queue is known to be short: behind the first-hit stage's compaction and from bounce 1 on.)
    __shared__ uint32_t s_n[2][kWfChunk / 64];
#pragma unroll 1
    for (int64_t chunk = blockIdx.x; chunk * kWfChunk < count; chunk += gridDim.x) {           // workgroup-uniform
    const int64_t q = chunk * kWfChunk + threadIdx.x;
    bool alive = false, shadow = false;
    // EPSM_TRACE_FUSE_FIRST_HIT: the primary rays' stage (packet kernel) has dealt with the paths that end at their first vertex and
    // the survivors come here compacted (kWfPreCompact); without that stage (EPSM_WF_NO_PACKET builds) this one does it, wf_shade's `out`
    const bool fuse = b == 0 && (A.flags & EPSM_TRACE_FUSE_FIRST_HIT) && !(A.flags & kWfPreCompact);     // (kernel-uniform)
    WfFirstHit fh;
    fh.rows.on = false; fh.rows.key[0] = fh.rows.key[1] = fh.rows.key[2] = kNoIndex;
    fh.rows.val[0] = fh.rows.val[1] = fh.rows.val[2] = fh.gd = zero3<float>();
    if (q < count) {
        wf_shade(A, W, wf_identity(A, b) ? q : (int64_t) W.queue[b & 1][q], b, alive, shadow, fuse ? &fh : nullptr);
        W.flags[q] = (uint8_t) ((alive ? kWfAlive : 0) | (shadow ? kWfShadow : 0));
    }
    if (fuse) first_hit_scatter(A, W, fh, (unsigned) chunk * (kWfChunk / 64) + (threadIdx.x >> 6));   // (all lanes of the wave: the sums run over it)
    const unsigned long long ma = __ballot(alive), ms = __ballot(shadow);
    if ((threadIdx.x & 63) == 0) { s_n[0][threadIdx.x >> 6] = (uint32_t) __popcll(ma); s_n[1][threadIdx.x >> 6] = (uint32_t) __popcll(ms); }
    __syncthreads();
    if (threadIdx.x < 2) {
        uint32_t n = 0;
#pragma unroll
        for (int w = 0; w < kWfChunk / 64; ++w) n += s_n[threadIdx.x][w];
        W.chunk_counts[threadIdx.x * W.chunks + chunk] = n;
        if (n) atomicAdd(&W.group_counts[threadIdx.x * W.groups + chunk / kWfGroup], n);     // (64 adds per address at most)
    }
    __syncthreads();                                                 // (the counts of this chunk are read before the next one writes its own)
    }
}
// Exclusive scan of the chunk counts of both queues (<= 65 536 chunks each at 2^24 paths), one workgroup PER QUEUE, two
// levels: the counts of the GROUPS of 64 chunks (summed by the shade stage) are scanned over the workgroup -- one coalesced
// load per thread at 2^24 paths --, then every wave takes groups in turn: one count per lane, a wave scan, the group's offset
// added.  Publishes the queue length of the next stage and leaves the group counts zeroed for the next bounce.
// (One level -- thread t sums a strip of 64 counts, the strips' sums are scanned, the strip is rewritten -- took 100 us for
// the first bounce of a 2^24-path tile: two passes of strided, dependent loads by two workgroups.)
__global__ __launch_bounds__(1024) void epsm_wf_scan_kernel(TraceArgs A, WfState W, int b) {
    const int64_t count = wf_count(A, W, b), n = (count + kWfChunk - 1) / kWfChunk, groups = (n + kWfGroup - 1) / kWfGroup;
    if (wf_in_tail(A, b, count)) return;                          // (the counters of the later bounces stay 0: their stages leave at once)
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    const int which = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t *c = W.chunk_counts + which * W.chunks, *gc = W.group_counts + which * W.groups, *go = W.group_offsets + which * W.groups;
    if (threadIdx.x == 0) s_carry = 0u;
    __syncthreads();
    for (int64_t base = 0; base < groups; base += 1024) {         // (one turn up to 2^24 paths)
        const int64_t g = base + threadIdx.x;
        const uint32_t v = g < groups ? gc[g] : 0u;
        if (g < groups) gc[g] = 0u;
        uint32_t inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t t = (uint32_t) __shfl_up((int) inc, off); if (lane >= off) inc += t; }
        if (lane == 63) s_w[wv] = inc;
        __syncthreads();
        uint32_t before = s_carry, all = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const uint32_t t = s_w[w]; if (w < wv) before += t; all += t; }
        if (g < groups) go[g] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) s_carry += all;
        __syncthreads();
    }
    if (threadIdx.x == 0) W.counters[which == 0 ? b + 1 : 8 + b] = s_carry;
    if (threadIdx.x == 0 && b < 0 && which == 0 && A.fh.survivor_count) *A.fh.survivor_count = s_carry;   // (the caller's copy: EpsmFirstHitBackward)
    constexpr int kBatch = 4;                                     // groups a wave has in flight
    for (int64_t g0 = (int64_t) wv * kBatch; g0 < groups; g0 += 16 * kBatch) {
        uint32_t v[kBatch], o[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const int64_t k = (g0 + j) * kWfGroup + lane;
            v[j] = (g0 + j < groups && k < n) ? c[k] : 0u;
            o[j] = g0 + j < groups ? go[g0 + j] : 0u;
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            uint32_t inc = v[j];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const uint32_t t = (uint32_t) __shfl_up((int) inc, off); if (lane >= off) inc += t; }
            const int64_t k = (g0 + j) * kWfGroup + lane;
            if (g0 + j < groups && k < n) c[k] = o[j] + inc - v[j];
        }
    }
}
// Writes the queue of bounce b + 1 and the shadow queue of bounce b, in path order (stable).
// (Tried: grouping the survivors of a chunk by the octant of their new direction, 8-bucket counting sort in LDS --
// 4.95 -> 5.85 ms at 128 k triangles, 8.1 -> 9.3 ms at 512 k: path order keeps the samples of a pixel, which start
// from almost the same point, next to each other, and that is worth more than a shared direction octant.)
// Round 4, EPSM_WF_REKEY=<shift>: inside its 256-slot chunk a queue is written grouped by the triangle the
// rays LEAVE (key = leaf-order triangle id >> shift hashed into 64 buckets, counting sort in LDS) instead of in path order:
// rays that leave one triangle (or one BVH leaf: ids are leaf-ordered) start together.  Different from the octant sort above:
// the chunk, and with it the pixel neighbourhood, is kept.
__global__ __launch_bounds__(kWfChunk) void epsm_wf_compact_kernel(TraceArgs A, WfState W, int b) {
    const int64_t count = wf_count(A, W, b);
    if (wf_in_tail(A, b, count)) return;
    for (int64_t chunk = blockIdx.x; chunk * kWfChunk < count; chunk += gridDim.x) {              // workgroup-uniform
    const int64_t q = chunk * kWfChunk + threadIdx.x;
    const uint8_t f = q < count ? W.flags[q] : (uint8_t) 0;
    const uint32_t i = q < count ? (wf_identity(A, b) ? (uint32_t) q : W.queue[b & 1][q]) : 0u;
#ifdef EPSM_WF_REKEY
    __shared__ uint32_t s_hist[2][64];
    if (threadIdx.x < 128) s_hist[threadIdx.x >> 6][threadIdx.x & 63] = 0u;
    const uint32_t tri = f ? W.hit[i].x : 0u;
    const uint32_t key = (((tri >> EPSM_WF_REKEY) * 2654435761u) >> 26) & 63u;
    __syncthreads();
    uint32_t rank[2] = {0u, 0u};
#pragma unroll
    for (int which = 0; which < 2; ++which)
        if ((f >> which) & 1) rank[which] = atomicAdd(&s_hist[which][key], 1u);
    __syncthreads();
    if (threadIdx.x < 128) {                                         // exclusive scan of the 64 buckets of both queues (one wave each)
        const int which = threadIdx.x >> 6, l = threadIdx.x & 63;
        const uint32_t c = s_hist[which][l];
        uint32_t inc = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t t = (uint32_t) __shfl_up((int) inc, off); if (l >= off) inc += t; }
        s_hist[which][l] = inc - c;
    }
    __syncthreads();
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        if (!((f >> which) & 1)) continue;
        const uint32_t off = W.chunk_counts[which * W.chunks + chunk] + s_hist[which][key] + rank[which];
        (which == 0 ? W.queue[(b + 1) & 1] : W.shadow_queue)[off] = i;
    }
#else
    __shared__ uint32_t s_n[2][kWfChunk / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long m[2] = {__ballot((f & kWfAlive) != 0), __ballot((f & kWfShadow) != 0)};
    if (lane == 0) { s_n[0][wv] = (uint32_t) __popcll(m[0]); s_n[1][wv] = (uint32_t) __popcll(m[1]); }
    __syncthreads();
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        if (!((f >> which) & 1)) continue;
        uint32_t off = W.chunk_counts[which * W.chunks + chunk];
        for (int w = 0; w < wv; ++w) off += s_n[which][w];
        off += (uint32_t) __popcll(m[which] & ((1ull << lane) - 1ull));
        (which == 0 ? W.queue[(b + 1) & 1] : W.shadow_queue)[off] = i;
        if (b < 0 && which == 0 && A.fh.survivors) A.fh.survivors[off] = i;      // (the caller's copy: EpsmFirstHitBackward)
    }
#endif
    __syncthreads();
    }
}
// EPSM_WF_QUAD (A/B build, round 5): the closest-hit stage of the bounces >= 1 with FOUR LANES PER RAY (epsm_trace_quad.h): a
// workgroup of 128 lanes takes 32 rays at a time.
__global__ __launch_bounds__(kWfThreads) void epsm_wf_extend_quad_kernel(TraceArgs A, WfState W, int b) {
    __shared__ uint32_t s_stack[kQuadStack * (kWfThreads / 4)];
    const int64_t count = wf_count(A, W, b);
    if (wf_in_tail(A, b, count)) return;
    const int lane = threadIdx.x & 63, quad = threadIdx.x >> 2;
    for (int64_t q0 = (int64_t) blockIdx.x * (kWfThreads / 4); q0 < count; q0 += (int64_t) gridDim.x * (kWfThreads / 4)) {
        const int64_t q = q0 + quad;
        const bool has = q < count;
        const int64_t i = has ? (int64_t) W.queue[b & 1][q] : 0;
        Ray r; r.o = r.d = zero3<float>(); r.maxt = 0.f;
        if (has) { const W4 o = W.ray_o[i], d = W.ray_d[i]; r.o = xyz(o); r.maxt = u2f(o.w); r.d = xyz(d); }
        const TriHit th = quad_intersect<false>(A.S, r, has, s_stack + quad, kWfThreads / 4, lane);
        if (has && (lane & 3) == 0) {
            W4 h; h.x = th.hit ? th.tri : kNoIndex; h.y = f2u(th.t); h.z = f2u(th.u); h.w = f2u(th.v);
            W.hit[i] = h;
        }
    }
}
__global__ __launch_bounds__(kWfThreads) void epsm_wf_shadow_kernel(TraceArgs A, WfState W, int b) {
    __shared__ uint32_t s_stack[kWfStackLds * kWfThreads];
    const int64_t count = (int64_t) W.counters[8 + b];
    for (int64_t q = (int64_t) blockIdx.x * kWfThreads + threadIdx.x; q < count; q += (int64_t) gridDim.x * kWfThreads)
        wf_shadow(A, W, (int64_t) W.shadow_queue[q], b, s_stack + threadIdx.x, kWfThreads);
}
__global__ __launch_bounds__(256) void epsm_wf_finish_kernel(TraceArgs A, WfState W) {
    for (int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x; i < A.N; i += (int64_t) gridDim.x * 256) wf_finish(A, W, i);
}

// ImageBlock::put (src/render/imageblock.cpp) with the reconstruction filter evaluated
// exactly (box: the pixel under the sample; gaussian: stddev 0.5, radius 4 sigma = 2,
// src/rfilters/gaussian.cpp) -- separable weights, float atomics into [r,g,b,w].
//
// The wavefront is ordered pixel-major, sample-minor (common.py:320-330), so the lanes of a wave are the
// samples of one pixel (spp >= 64) or of a few pixels (8, 16, 32 spp).  Lanes of an aligned group of G
// lanes that sit in the same pixel splat onto the same 5x5 window: their weighted radiances are summed
// with DPP adds and ONE lane issues the four atomics of a window pixel -- 100 atomics per group instead
// of 64 x 16 x 4 per wave (the render was bound by the atomic rate: 68 ms at 512x512 @ 64 spp).
// Arbitrary sample positions are still correct: a wave whose groups are not uniform splats lane by lane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float splat_dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
// Sum over aligned groups of G lanes (all 64 lanes active); the last lane of each group holds the total.
template <int G> __device__ __forceinline__ float group_total(float v) {
    v = splat_dpp_add<0xB1, 0xf>(v);                    // quad_perm [1,0,3,2]
    v = splat_dpp_add<0x4E, 0xf>(v);                    // quad_perm [2,3,0,1]
    v = splat_dpp_add<0x141, 0xf>(v);                   // row_half_mirror: 8 lanes
    if (G >= 16) v = splat_dpp_add<0x140, 0xf>(v);      // row_mirror: 16 lanes
    if (G == 64) {
        v = splat_dpp_add<0x142, 0xa>(v);               // row_bcast15
        v = splat_dpp_add<0x143, 0xc>(v);               // row_bcast31
    }
    return v;
}
struct SplatWeights { float wx[5], wy[5]; int X, Y; };
__device__ __forceinline__ SplatWeights gaussian_window(float px, float py) {
    SplatWeights w;
    w.X = (int) floorf(px); w.Y = (int) floorf(py);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const float dx = (w.X + j - 2 + 0.5f) - px, dy = (w.Y + j - 2 + 0.5f) - py;
        // pixel centres farther than the radius get weight 0: the 5x5 window holds
        // exactly the pixels ImageBlock::put visits (ceil(p - r - 0.5) .. floor(p + r - 0.5))
        w.wx[j] = gaussian_rfilter(dx);
        w.wy[j] = gaussian_rfilter(dy);
    }
    return w;
}
template <int G>
__device__ __forceinline__ void splat_groups(const SplatWeights &w, float r, float g, float b, int W, int H, float *accum) {
    const bool carrier = (threadIdx.x & (G - 1)) == G - 1;
#pragma unroll
    for (int jy = 0; jy < 5; ++jy) {
        const int y = w.Y + jy - 2;
#pragma unroll
        for (int jx = 0; jx < 5; ++jx) {
            const int x = w.X + jx - 2;
            const float wt = w.wy[jy] * w.wx[jx];
            const float sr = group_total<G>(r * wt), sg = group_total<G>(g * wt), sb = group_total<G>(b * wt),
                        sw = group_total<G>(wt);
            if (carrier && sw != 0.f && x >= 0 && y >= 0 && x < W && y < H) {
                float *a = accum + 4 * ((int64_t) y * W + x);
                atomicAdd(a + 0, sr); atomicAdd(a + 1, sg); atomicAdd(a + 2, sb); atomicAdd(a + 3, sw);
            }
        }
    }
}
__global__ __launch_bounds__(256) void epsm_film_splat_kernel(int64_t N, const float *pos, const float *rad, int W, int H,
                                                              int rfilter, float *accum) {
    const int64_t i0 = (int64_t) blockIdx.x * 256 + threadIdx.x;
    const bool live = i0 < N;
    const int64_t i = live ? i0 : N - 1;                // lanes past the end take part in the sums with weight 0
    const float px = pos[2 * i], py = pos[2 * i + 1];
    float r = rad[3 * i], g = rad[3 * i + 1], b = rad[3 * i + 2];
    if (rfilter == EPSM_RFILTER_BOX) {
        const int x = (int) floorf(px), y = (int) floorf(py);
        if (!live || x < 0 || y < 0 || x >= W || y >= H) return;
        float *a = accum + 4 * ((int64_t) y * W + x);
        atomicAdd(a + 0, r); atomicAdd(a + 1, g); atomicAdd(a + 2, b); atomicAdd(a + 3, 1.f);
        return;
    }
    SplatWeights w = gaussian_window(px, py);
    if (!live) {
#pragma unroll
        for (int j = 0; j < 5; ++j) w.wx[j] = w.wy[j] = 0.f;
    }
    // largest aligned group size whose lanes all sit in one pixel, the same for the whole wave
    const int lane = threadIdx.x & 63;
    const int X8 = __shfl(w.X, lane & ~7), Y8 = __shfl(w.Y, lane & ~7);
    const int X16 = __shfl(w.X, lane & ~15), Y16 = __shfl(w.Y, lane & ~15);
    const int X64 = __builtin_amdgcn_readfirstlane(w.X), Y64 = __builtin_amdgcn_readfirstlane(w.Y);
    if (__ballot(w.X != X64 || w.Y != Y64) == 0ull) { splat_groups<64>(w, r, g, b, W, H, accum); return; }
    if (__ballot(w.X != X16 || w.Y != Y16) == 0ull) { splat_groups<16>(w, r, g, b, W, H, accum); return; }
    if (__ballot(w.X != X8 || w.Y != Y8) == 0ull) { splat_groups<8>(w, r, g, b, W, H, accum); return; }
#pragma unroll
    for (int jy = 0; jy < 5; ++jy) {
        const int y = w.Y + jy - 2;
        if (y < 0 || y >= H) continue;
#pragma unroll
        for (int jx = 0; jx < 5; ++jx) {
            const int x = w.X + jx - 2;
            const float wt = w.wy[jy] * w.wx[jx];
            if (x < 0 || x >= W || wt == 0.f) continue;
            float *a = accum + 4 * ((int64_t) y * W + x);
            atomicAdd(a + 0, r * wt); atomicAdd(a + 1, g * wt); atomicAdd(a + 2, b * wt); atomicAdd(a + 3, wt);
        }
    }
}
// Adjoint of the gaussian splat + weight division w.r.t. a sample's radiance, its FILM POSITION and the determinant of the
// reparameterisation that multiplies both its value and its weight (common.py:880-920; prb_reparam's pass between its two
// traces):   image[p] = sum_i w_ip L_i det_i / sum_i w_ip det_i,   w_ip = f(p - pos_i).
// One lane per sample, its 5 x 5 window of pixels: dL = sum_p w_ip grad_p / W_p;  A_p = grad_p . (L_i - image_p) / W_p;
// d/d pos.x = sum_p A_p wy dwx,  d/d pos.y = sum_p A_p dwy wx,  d/d det = sum_p A_p w.  (Round 3 ran this as ~40 small torch
// kernels per tile: integrators.film_adjoint_reparam, kept as the checker.)
__global__ __launch_bounds__(256) void epsm_film_adjoint_reparam_kernel(int64_t N, const float *pos, const float *rad, const float *grad,
                                                                        int grad_stride, const float *accum, int W, int H,
                                                                        float *dL, float *adj) {
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float px = pos[2 * i], py = pos[2 * i + 1];
    const float r = rad[3 * i], g = rad[3 * i + 1], b = rad[3 * i + 2];
    const float radius = 2.f, alpha = -1.f / (2.f * 0.5f * 0.5f), bias = expf(alpha * radius * radius);
    const int X = (int) floorf(px), Y = (int) floorf(py);
    float wx[5], wy[5], dwx[5], dwy[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int x = X + j - 2, y = Y + j - 2;
        const float dx = (x + 0.5f) - px, dy = (y + 0.5f) - py;
        const float ex = expf(alpha * dx * dx), ey = expf(alpha * dy * dy);
        const bool lx = fabsf(dx) <= radius && x >= 0 && x < W && ex - bias > 0.f;
        const bool ly = fabsf(dy) <= radius && y >= 0 && y < H && ey - bias > 0.f;
        wx[j] = lx ? ex - bias : 0.f; dwx[j] = lx ? -2.f * alpha * dx * ex : 0.f;      // d w / d pos (dx = pixel centre - pos)
        wy[j] = ly ? ey - bias : 0.f; dwy[j] = ly ? -2.f * alpha * dy * ey : 0.f;
    }
    float lr = 0.f, lg = 0.f, lb = 0.f, ax = 0.f, ay = 0.f, ad = 0.f;
#pragma unroll
    for (int jy = 0; jy < 5; ++jy) {
        const int y = min(max(Y + jy - 2, 0), H - 1);
#pragma unroll
        for (int jx = 0; jx < 5; ++jx) {
            const int x = min(max(X + jx - 2, 0), W - 1);
            const float w2 = wy[jy] * wx[jx], wdx = wy[jy] * dwx[jx], wdy = dwy[jy] * wx[jx];
            if (w2 == 0.f && wdx == 0.f && wdy == 0.f) continue;
            const int64_t p = (int64_t) y * W + x;
            const float Wp = accum[4 * p + 3];
            const float inv = Wp > 0.f ? 1.f / fmaxf(Wp, 1e-30f) : 0.f;
            const float *gp = grad + p * grad_stride;
            const float gr = gp[0] * inv, gg = gp[1] * inv, gb = gp[2] * inv;             // grad / W_p
            const float gi = (gr * accum[4 * p] + gg * accum[4 * p + 1] + gb * accum[4 * p + 2]) * inv;   // (grad . image) / W_p
            const float A = gr * r + gg * g + gb * b - gi;
            lr += gr * w2; lg += gg * w2; lb += gb * w2;
            ax += A * wdx; ay += A * wdy; ad += A * w2;
        }
    }
    dL[3 * i] = lr; dL[3 * i + 1] = lg; dL[3 * i + 2] = lb;
    adj[3 * i] = ax; adj[3 * i + 1] = ay; adj[3 * i + 2] = ad;
}
__global__ __launch_bounds__(256) void epsm_film_develop_kernel(int64_t n, const float *accum, float *image) {
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float w = accum[4 * i + 3], iw = w != 0.f ? 1.f / w : 0.f;
    image[3 * i] = accum[4 * i] * iw; image[3 * i + 1] = accum[4 * i + 1] * iw; image[3 * i + 2] = accum[4 * i + 2] * iw;
}

}  // namespace

// Validates the arguments of the two tracer entry points and fills the kernel argument block.
static int fill_trace_args(TraceArgs &A, const char *who, const EpsmScene *scene, const EpsmSensor *sensor,
                           uint32_t seed, int spp, int max_depth, int rr_depth, int64_t path_offset, int64_t N, int K_log,
                           float *ray_o, float *ray_d, float *ray_dx, float *ray_dy, float *film_pos, float *radiance,
                           uint8_t *valid, const EpsmRecordOut *recs, uint32_t flags) {
    char msg[200];
    auto bad = [&](const char *what) { snprintf(msg, sizeof(msg), "%s: %s", who, what); return fail(EPSM_EINVAL, msg); };
    if (!scene || !sensor) return bad("NULL scene / sensor");
    if (N < 0 || spp < 1 || max_depth < 1 || rr_depth < 1 || path_offset < 0)
        return bad("bad N / spp / max_depth / rr_depth / path_offset");
    if (K_log < 0 || K_log > EPSM_MAX_VERTICES || K_log > (max_depth < 6 ? max_depth : 6))
        return bad("K_log must be <= min(max_depth, 5)");
    if (sensor->border < 0 || sensor->border > 8) return bad("bad sensor border");
    if (path_offset + N > (int64_t) (sensor->width + 2 * sensor->border) * (sensor->height + 2 * sensor->border) * spp ||
        path_offset + N > 0xFFFFFFFFLL)
        return bad("path range exceeds (width + 2 border) * (height + 2 border) * spp (or 2^32, common.py:468-475)");
    const bool packed = (flags & EPSM_TRACE_PACKED_LOG) != 0;
    if (!ray_o || (!packed && (!ray_d || !ray_dx || !ray_dy)) || (K_log > 0 && !recs)) return bad("NULL output");
    if (packed && ((((uintptr_t) ray_o) & 15) || (K_log > 0 && (!recs[0].packed || !recs[0].pflags || (((uintptr_t) recs[0].packed) & 15)))))
        return bad("EPSM_TRACE_PACKED_LOG needs 16-byte aligned ray_o (N,12), recs[0].packed and recs[0].pflags");
    if (packed && K_log > 0) {
        const int64_t rs = recs[0].ray_stride, ps = recs[0].packed_stride;
        if (rs < 0 || ps < 0 || (rs & 3) || (ps & 3) || (rs && rs < 12) || (ps && ps < (int64_t) K_log * 32))
            return bad("EpsmRecordOut.ray_stride / packed_stride must be 0 or multiples of 4 words >= 12 / 32 K_log");
    }
    if (scene->n_triangles > 0 && (!scene->positions || !scene->normals || !scene->tri || !scene->tri_mesh ||
                                   !scene->meshes || !scene->bsdfs || !scene->bvh || !scene->prim_index || !scene->tri_verts))
        return bad("NULL scene array");
    if (const char *why = epsm_host::scene_tables_invalid(scene)) return bad(why);
    if (flags & ~(uint32_t) (EPSM_TRACE_SPARSE_LOG | EPSM_TRACE_PACKED_LOG | EPSM_TRACE_GRADIENT_ONLY | EPSM_TRACE_GRADIENT_CAUSTIC | EPSM_TRACE_NO_TAIL |
                             EPSM_TRACE_FUSE_FIRST_HIT))
        return bad("unknown flag");
    if (flags & EPSM_TRACE_FUSE_FIRST_HIT) {
        if (!(flags & EPSM_TRACE_GRADIENT_ONLY) || !packed || K_log < 1) return bad("EPSM_TRACE_FUSE_FIRST_HIT needs EPSM_TRACE_GRADIENT_ONLY and EPSM_TRACE_PACKED_LOG");
        const EpsmFirstHitBackward *f = recs[0].first_hit;
        if (!f || !f->grad_img || !f->grad_pos || (f->T > 0 && !f->tri_table) || f->T < 0 || f->V < 0 || f->res < 1 || f->img_width < f->res ||
            f->img_channels < 5 || (((uintptr_t) f->tri_table) & 15))
            return bad("EPSM_TRACE_FUSE_FIRST_HIT: recs[0].first_hit needs grad_img (>= 5 channels, img_width >= res >= 1), grad_pos and a 16-byte aligned tri_table");
        if (path_offset + N > (int64_t) f->res * f->res * spp) return bad("EPSM_TRACE_FUSE_FIRST_HIT: path range exceeds res * res * spp");
        if (recs[0].shadow && max_depth <= 3) return bad("EPSM_TRACE_FUSE_FIRST_HIT: not with the occluder record (max_depth <= 3)");
        if ((f->survivors != nullptr) != (f->survivor_count != nullptr)) return bad("EPSM_TRACE_FUSE_FIRST_HIT: survivors and survivor_count come together");
    }
    if ((flags & EPSM_TRACE_GRADIENT_CAUSTIC) && !(flags & EPSM_TRACE_GRADIENT_ONLY)) return bad("EPSM_TRACE_GRADIENT_CAUSTIC modifies EPSM_TRACE_GRADIENT_ONLY");
    if ((flags & EPSM_TRACE_GRADIENT_ONLY) && K_log < 1) return bad("EPSM_TRACE_GRADIENT_ONLY needs a vertex log (K_log >= 1)");
    memset(&A, 0, sizeof(A));
    A.flags = flags;
    A.S = *scene; A.C = *sensor;
    A.seed = seed; A.spp = spp; A.max_depth = max_depth; A.rr_depth = rr_depth; A.K_log = K_log;
    A.path_offset = path_offset; A.N = N;
    A.ray_o = ray_o; A.ray_d = ray_d; A.ray_dx = ray_dx; A.ray_dy = ray_dy;
    A.film_pos = film_pos; A.radiance = radiance; A.valid = valid;
    for (int k = 0; k < K_log; ++k) {
        const EpsmRecordOut &r = recs[k];
        if (packed) { A.rec[k] = r; continue; }
        if (!r.p0 || !r.p1 || !r.p2 || !r.n0 || !r.n1 || !r.n2 || !r.b0 || !r.b1 || !r.eta || !r.hf || !r.light ||
            !r.bsdf || !r.active || !r.active_em || !r.ismesh || !r.tri || !r.aux || !r.emit)
            return bad("NULL pointer in a record (p / normal may be NULL)");
        A.rec[k] = r;
    }
    trace_args_log_strides(A);
    if (flags & EPSM_TRACE_FUSE_FIRST_HIT) A.fh = *recs[0].first_hit;
    return EPSM_OK;
}

extern "C" int epsm_trace_paths(const EpsmScene *scene, const EpsmSensor *sensor,
                                uint32_t seed, int spp, int max_depth, int rr_depth,
                                int64_t path_offset, int64_t N, int K_log,
                                float *ray_o, float *ray_d, float *ray_dx, float *ray_dy,
                                float *film_pos, float *radiance, uint8_t *valid,
                                const EpsmRecordOut *recs, uint32_t flags, void *stream) {
    epsm_host::err_buf()[0] = 0;
    if (scene && sensor && N == 0) return EPSM_OK;
    TraceArgs A;
    if (flags & EPSM_TRACE_FUSE_FIRST_HIT) return fail(EPSM_EINVAL, "epsm_trace_paths: EPSM_TRACE_FUSE_FIRST_HIT is the wavefront form's (epsm_trace_paths_wavefront)");
    const int rc = fill_trace_args(A, "epsm_trace_paths", scene, sensor, seed, spp, max_depth, rr_depth, path_offset, N, K_log,
                                   ray_o, ray_d, ray_dx, ray_dy, film_pos, radiance, valid, recs, flags);
    if (rc != EPSM_OK) return rc;
    hipLaunchKernelGGL(epsm_trace_kernel, dim3((unsigned) ((N + 127) / 128)), dim3(128), 0, (hipStream_t) stream, A);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return epsm_host::hip_fail("epsm_trace_paths", e);
    return EPSM_OK;
}

extern "C" int epsm_trace_paths_color(const EpsmScene *scene, const EpsmSensor *sensor,
                                      uint32_t seed, int spp, int max_depth, int rr_depth,
                                      int64_t path_offset, int64_t N,
                                      float *film_pos, float *radiance, uint8_t *valid,
                                      float *color_sum, int n_color, void *stream) {
    epsm_host::err_buf()[0] = 0;
    if (scene && sensor && N == 0) return EPSM_OK;
    if (!color_sum || n_color < 1 || n_color > 4 || !radiance)
        return fail(EPSM_EINVAL, "epsm_trace_paths_color: need color_sum, radiance and 1 <= n_color <= 4");
    TraceArgs A;
    static float dummy_rays[12];                               // (the rays are not wanted: fill_trace_args only checks for NULL)
    const int rc = fill_trace_args(A, "epsm_trace_paths_color", scene, sensor, seed, spp, max_depth, rr_depth, path_offset, N, 0,
                                   dummy_rays, dummy_rays, dummy_rays, dummy_rays, film_pos, radiance, valid, nullptr, 0);
    if (rc != EPSM_OK) return rc;
    A.ray_o = A.ray_d = A.ray_dx = A.ray_dy = nullptr;
    A.color_sum = color_sum; A.n_color = n_color;
    hipError_t e = hipMemsetAsync(color_sum, 0, (size_t) N * n_color * 3 * sizeof(float), (hipStream_t) stream);
    if (e != hipSuccess) return epsm_host::hip_fail("epsm_trace_paths_color", e);
    hipLaunchKernelGGL(epsm_trace_kernel, dim3((unsigned) ((N + 127) / 128)), dim3(128), 0, (hipStream_t) stream, A);
    e = hipGetLastError();
    if (e != hipSuccess) return epsm_host::hip_fail("epsm_trace_paths_color", e);
    return EPSM_OK;
}

extern "C" size_t epsm_trace_workspace_bytes(int64_t N) { return N > 0 ? wf_workspace_bytes(N) : 0; }

extern "C" int epsm_trace_paths_wavefront(const EpsmScene *scene, const EpsmSensor *sensor,
                                          uint32_t seed, int spp, int max_depth, int rr_depth,
                                          int64_t path_offset, int64_t N, int K_log,
                                          float *ray_o, float *ray_d, float *ray_dx, float *ray_dy,
                                          float *film_pos, float *radiance, uint8_t *valid,
                                          const EpsmRecordOut *recs, uint32_t flags, void *workspace, size_t workspace_bytes, void *stream) {
    epsm_host::err_buf()[0] = 0;
    if (scene && sensor && N == 0) return EPSM_OK;
    TraceArgs A;
    const int rc = fill_trace_args(A, "epsm_trace_paths_wavefront", scene, sensor, seed, spp, max_depth, rr_depth, path_offset,
                                   N, K_log, ray_o, ray_d, ray_dx, ray_dy, film_pos, radiance, valid, recs, flags);
    if (rc != EPSM_OK) return rc;
    if (!workspace || (((uintptr_t) workspace) & 15) || workspace_bytes < wf_workspace_bytes(N))
        return fail(EPSM_EINVAL, "epsm_trace_paths_wavefront: workspace NULL, not 16-byte aligned or smaller than epsm_trace_workspace_bytes(N)");
    if (N > 0xFFFFFFF0LL) return fail(EPSM_EINVAL, "epsm_trace_paths_wavefront: N must fit the 32-bit queues");
    hipStream_t s = (hipStream_t) stream;
    const WfState W = wf_carve(workspace, N);
    hipError_t e = hipMemsetAsync(W.counters, 0, wf_zeroed_bytes(N), s);          // the counters and the group counts behind them
    if (e != hipSuccess) return epsm_host::hip_fail("epsm_trace_paths_wavefront", e);
#ifdef EPSM_WF_NO_PACKET
    if ((A.flags & EPSM_TRACE_FUSE_FIRST_HIT) && A.fh.survivors)
        return fail(EPSM_EINVAL, "epsm_trace_paths_wavefront: this build (EPSM_WF_NO_PACKET) compacts no survivors: first_hit->survivors must be NULL");
#else
    if (A.flags & EPSM_TRACE_FUSE_FIRST_HIT) {
        A.flags |= kWfPreCompact;                                                  // (the packet stage ADDS its survivors to the chunk counts)
        e = hipMemsetAsync(W.chunk_counts, 0, (size_t) W.chunks * 8, s);
        if (e != hipSuccess) return epsm_host::hip_fail("epsm_trace_paths_wavefront", e);
    }
#endif
    auto blocks = [&](int threads) { const int64_t b = (N + threads - 1) / threads; return dim3((unsigned) (b < kWfMaxBlocks ? b : kWfMaxBlocks)); };
    const int depth = path_max_depth(A);
    const dim3 shade_short((unsigned) (W.chunks < 16384 ? W.chunks : 16384));
    const dim3 chunks((unsigned) W.chunks), chunk_blocks((unsigned) (W.chunks < kWfMaxChunkBlocks ? W.chunks : kWfMaxChunkBlocks));
    const int64_t tail_most = N < kWfTailBelow ? N : kWfTailBelow;
    const dim3 tail_blocks((unsigned) ((tail_most + kWfThreads - 1) / kWfThreads));
    for (int b = 0; b < depth; ++b) {
        // the queue lengths of bounce b live on the device: every stage is launched for the worst case and its
        // surplus workgroups leave at once (no host round trip between the bounces)
        // (bounces >= 1: the tail first -- it runs once, when the queue has become short, and the stages behind it leave at once)
        if (b >= 1 && !(A.flags & EPSM_TRACE_NO_TAIL)) hipLaunchKernelGGL(epsm_wf_tail_kernel, tail_blocks, dim3(kWfThreads), 0, s, A, W, b);
        if (b == 0) {
#ifdef EPSM_WF_NO_PACKET
            hipLaunchKernelGGL(epsm_wf_extend_kernel<true>, blocks(kWfThreads), dim3(kWfThreads), 0, s, A, W, b);
#else
            hipLaunchKernelGGL(epsm_wf_extend_packet_kernel<true>, blocks(kWfThreads), dim3(kWfThreads), 0, s, A, W, b);
            if (A.flags & kWfPreCompact) {                       // the survivors of the primary rays' stage, compacted: "bounce -1"
                hipLaunchKernelGGL(epsm_wf_scan_kernel, dim3(2), dim3(1024), 0, s, A, W, -1);
                hipLaunchKernelGGL(epsm_wf_compact_kernel, chunk_blocks, dim3(kWfChunk), 0, s, A, W, -1);
            }
#endif
        }
#if defined(EPSM_WF_QUAD)
        else hipLaunchKernelGGL(epsm_wf_extend_quad_kernel, blocks(kWfThreads / 4), dim3(kWfThreads), 0, s, A, W, b);
#elif defined(EPSM_WF_PACKET_BOUNCE)
        else hipLaunchKernelGGL(epsm_wf_extend_packet_kernel<false>, blocks(kWfThreads), dim3(kWfThreads), 0, s, A, W, b);
#else
        else hipLaunchKernelGGL(epsm_wf_extend_kernel<false>, blocks(kWfThreads), dim3(kWfThreads), 0, s, A, W, b);
#endif
        // (the whole grid where every chunk has work -- bounce 0 of an unfused trace --, a capped one where the queue is short)
        hipLaunchKernelGGL(epsm_wf_shade_kernel, (b == 0 && !(A.flags & kWfPreCompact)) ? chunks : shade_short, dim3(kWfChunk), 0, s, A, W, b);
        if (b == 0 && (A.flags & EPSM_TRACE_FUSE_FIRST_HIT) && A.fh.grad_o_sum)
            hipLaunchKernelGGL(epsm_wf_first_hit_finish_kernel, dim3(1), dim3(kWfFirstHitSlots), 0, s, A, W);
        hipLaunchKernelGGL(epsm_wf_scan_kernel, dim3(2), dim3(1024), 0, s, A, W, b);
        hipLaunchKernelGGL(epsm_wf_compact_kernel, chunk_blocks, dim3(kWfChunk), 0, s, A, W, b);
#ifdef EPSM_WF_PACKET_SHADOW
        hipLaunchKernelGGL(epsm_wf_shadow_packet_kernel, blocks(kWfThreads), dim3(kWfThreads), 0, s, A, W, b);
#else
        hipLaunchKernelGGL(epsm_wf_shadow_kernel, blocks(kWfThreads), dim3(kWfThreads), 0, s, A, W, b);
#endif
    }
    // (radiance / valid not asked for and the native log: nothing is left to write -- the gradient-only trace of render_backward)
    if (A.radiance || A.valid || !(A.flags & EPSM_TRACE_PACKED_LOG))
        hipLaunchKernelGGL(epsm_wf_finish_kernel, blocks(256), dim3(256), 0, s, A, W);
    e = hipGetLastError();
    if (e != hipSuccess) return epsm_host::hip_fail("epsm_trace_paths_wavefront", e);
    return EPSM_OK;
}

extern "C" int epsm_film_splat(int64_t N, const float *film_pos, const float *radiance, int width, int height,
                               int rfilter, float *accum, void *stream) {
    epsm_host::err_buf()[0] = 0;
    if (N == 0) return EPSM_OK;
    if (N < 0 || !film_pos || !radiance || !accum || width < 1 || height < 1)
        return fail(EPSM_EINVAL, "epsm_film_splat: bad argument");
    hipLaunchKernelGGL(epsm_film_splat_kernel, dim3((unsigned) ((N + 255) / 256)), dim3(256), 0, (hipStream_t) stream,
                       N, film_pos, radiance, width, height, rfilter, accum);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return epsm_host::hip_fail("epsm_film_splat", e);
    return EPSM_OK;
}
extern "C" int epsm_film_adjoint_reparam(int64_t N, const float *film_pos, const float *radiance, const float *grad_img,
                                         int grad_channels, const float *accum, int width, int height, float *dL, float *adj,
                                         void *stream) {
    epsm_host::err_buf()[0] = 0;
    if (N == 0) return EPSM_OK;
    if (N < 0 || !film_pos || !radiance || !grad_img || !accum || !dL || !adj || width < 1 || height < 1 || grad_channels < 3)
        return fail(EPSM_EINVAL, "epsm_film_adjoint_reparam: bad argument");
    hipLaunchKernelGGL(epsm_film_adjoint_reparam_kernel, dim3((unsigned) ((N + 255) / 256)), dim3(256), 0, (hipStream_t) stream,
                       N, film_pos, radiance, grad_img, grad_channels, accum, width, height, dL, adj);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return epsm_host::hip_fail("epsm_film_adjoint_reparam", e);
    return EPSM_OK;
}
extern "C" int epsm_film_develop(int width, int height, const float *accum, float *image, void *stream) {
    epsm_host::err_buf()[0] = 0;
    if (!accum || !image || width < 1 || height < 1) return fail(EPSM_EINVAL, "epsm_film_develop: bad argument");
    const int64_t n = (int64_t) width * height;
    hipLaunchKernelGGL(epsm_film_develop_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, (hipStream_t) stream,
                       n, accum, image);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return epsm_host::hip_fail("epsm_film_develop", e);
    return EPSM_OK;
}
