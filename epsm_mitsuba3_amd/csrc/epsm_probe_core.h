// epsm_probe_core.h -- one row of epsm_probe (include/epsm_trace.h): evaluates ONE of the tracer's per-path functions on
// plain numbers, so that the known answers the reference's own tests hold for them (tests/golden/reference_vectors.py)
// can be checked against the very code the tracer runs -- on the device through the product library, on the host through
// tests/host_harness/trace_host.cpp.  Rows are EPSM_PROBE_IN floats in, EPSM_PROBE_OUT floats out; integers travel as bits.
#pragma once

#include "epsm_trace_core.h"

namespace epsm {


EPSM_HD void probe_row(int what, const float *in, float *out, const EpsmBsdf *bsdf, const EpsmSensor *sensor) {
    for (int j = 0; j < EPSM_PROBE_OUT; ++j) out[j] = 0.f;
    switch (what) {
        case EPSM_PROBE_TEA: {                         // in: v0, v1 (bits) -> out: v0', v1' (bits)
            uint32_t o0, o1;
            sample_tea_32(f2u(in[0]), f2u(in[1]), o0, o1);
            out[0] = u2f(o0); out[1] = u2f(o1);
        } break;
        case EPSM_PROBE_PCG32: {                       // in: initstate lo, hi, initseq lo, hi (bits)
            Pcg32 r = pcg32_seed((uint64_t) f2u(in[0]) | ((uint64_t) f2u(in[1]) << 32),
                                 (uint64_t) f2u(in[2]) | ((uint64_t) f2u(in[3]) << 32));
            for (int j = 0; j < 6; ++j) out[j] = u2f(r.next_u32());       // six draws as bits
            for (int j = 6; j < 12; ++j) out[j] = r.next_1d();            // the next six as next_1d()
        } break;
        case EPSM_PROBE_SAMPLER: {                     // in: seed, wavefront index (bits): the tracer's per-path stream
            Pcg32 r = seed_sampler(f2u(in[0]), f2u(in[1]));
            for (int j = 0; j < 12; ++j) out[j] = r.next_1d();
        } break;
        case EPSM_PROBE_MICROFACET: {                  // in: m (3), wi (3) -> D(m), pdf(wi, m), G1(m; wi)
            const F3 m = f3(in[0], in[1], in[2]), wi = f3(in[3], in[4], in[5]);
            out[0] = mf_eval(*bsdf, m); out[1] = mf_pdf(*bsdf, wi, m); out[2] = mf_smith_g1(*bsdf, m, wi);
        } break;
        case EPSM_PROBE_MICROFACET_SAMPLE: {           // in: u1, u2 -> m (3), pdf, d m / d alpha (3)
            float pdf; F3 dm;
            const F3 m = mf_sample(*bsdf, in[0], in[1], pdf, dm);
            out[0] = m.x; out[1] = m.y; out[2] = m.z; out[3] = pdf; out[4] = dm.x; out[5] = dm.y; out[6] = dm.z;
        } break;
        case EPSM_PROBE_FRESNEL: {                     // in: cos_theta_i, eta -> F, cos_theta_t, eta_it, eta_ti
            fresnel(in[0], in[1], out[0], out[1], out[2], out[3]);
        } break;
        case EPSM_PROBE_FRESNEL_CONDUCTOR: {           // in: cos_theta_i, eta, k -> F
            out[0] = fresnel_conductor(in[0], in[1], in[2]);
        } break;
        case EPSM_PROBE_RFILTER: {                     // in: x -> gaussian(x)
            out[0] = gaussian_rfilter(in[0]);
        } break;
        case EPSM_PROBE_PRIMARY_RAY: {                 // in: film position (pixels) -> o, d, d_x, d_y
            const PrimaryRay p = primary_ray_at(*sensor, in[0], in[1]);
            out[0] = p.ray.o.x; out[1] = p.ray.o.y; out[2] = p.ray.o.z; out[3] = p.ray.d.x; out[4] = p.ray.d.y; out[5] = p.ray.d.z;
            out[6] = p.dx.x; out[7] = p.dx.y; out[8] = p.dx.z; out[9] = p.dy.x; out[10] = p.dy.y; out[11] = p.dy.z;
        } break;
        case EPSM_PROBE_BSDF_SAMPLE: {                 // in: wi (3), sample1, sample2 (2) -> wo, weight, pdf, eta, sampled_type, valid
            const BsdfSample b = bsdf_sample(*bsdf, f3(in[0], in[1], in[2]), in[3], in[4], in[5], true);
            out[0] = b.wo.x; out[1] = b.wo.y; out[2] = b.wo.z; out[3] = b.weight.x; out[4] = b.weight.y; out[5] = b.weight.z;
            out[6] = b.pdf; out[7] = b.eta; out[8] = u2f(b.sampled_type); out[9] = b.valid ? 1.f : 0.f;
        } break;
        case EPSM_PROBE_BSDF_EVAL: {                   // in: wi (3), wo (3) -> value incl. the cosine (3), pdf
            F3 v; float pdf;
            bsdf_eval_pdf(*bsdf, f3(in[0], in[1], in[2]), f3(in[3], in[4], in[5]), v, pdf);
            out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = pdf;
        } break;
        default: break;
    }
}

EPSM_HD bool probe_needs_bsdf(int what) {
    return what == EPSM_PROBE_MICROFACET || what == EPSM_PROBE_MICROFACET_SAMPLE || what == EPSM_PROBE_BSDF_SAMPLE || what == EPSM_PROBE_BSDF_EVAL;
}
EPSM_HD bool probe_needs_sensor(int what) { return what == EPSM_PROBE_PRIMARY_RAY; }

}  // namespace epsm
