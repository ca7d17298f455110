"""Known answers that the reference's OWN unit tests hold for the pieces of the hot path's surroundings that cannot be
imported here without Dr.Jit / Mitsuba (first-vertex tangent, scatter, sampler seeding, microfacet / Fresnel / filter
code of the tracer, camera ray differentials).  VALUES only -- inputs and expected outputs -- each with the place in
/root/reference it was read from; no reference source text.  Consumed by tests/test_reference_vectors.py (oracle + host
builds) and tests/test_gpu_reference_vectors.py (the device code through the C ABI).
"""
import math

import numpy as np

# ----------------------------------------------------------------------------------------------------------------------
# Tiny Encryption Algorithm, 4 rounds.  src/core/tests/test_random.py:9-16 (float32) and :20-27 (float64);
# sample_tea_float32 = bits((v1' >> 9) | 0x3f800000) - 1, sample_tea_float64 = bits(((v0' + (v1' << 32)) >> 12) | 0x3ff0...) - 1
# (include/mitsuba/core/random.h:111-114, 137-139, 161-163).
TEA_INPUTS = [(1, 1), (1, 2), (1, 3), (1, 4), (1, 5), (2, 1), (3, 1), (4, 1)]
TEA_FLOAT32 = [0.5424730777740479, 0.5079904794692993, 0.4171961545944214, 0.008385419845581055,
               0.8085528612136841, 0.6939879655838013, 0.6978365182876587, 0.4897364377975464]
TEA_FLOAT64 = [0.5424730799533735, 0.5079905082233922, 0.4171962610608142, 0.008385529523330604,
               0.80855288317879, 0.6939880404156831, 0.6978365636630994, 0.48973647949223253]

# PCG32.  src/samplers/tests/test_independent.py:16-28 states the identity "independent sampler == the PCG32 stream, consumed
# in order: next_1d takes one draw, next_2d two"; Dr.Jit (ext/drjit, an EMPTY submodule here, pinned 0.4.0 by pyproject.toml:2)
# implements M. O'Neill's pcg32 whose published check values (pcg-c-basic's pcg32-demo, seed 42 / stream 54) are:
PCG32_DEMO_SEED = (42, 54)
PCG32_DEMO_OUTPUT = [0xa15c02b7, 0x7b47f409, 0xba1d3330, 0x83d2f293, 0xbfa4784b, 0xcbed606e]
PCG32_DEFAULT_STATE, PCG32_DEFAULT_STREAM, PCG32_MULT = 0x853c49e6748fea9b, 0xda3e39cb94b95bdb, 0x5851f42d4c957f2d

# ----------------------------------------------------------------------------------------------------------------------
# rectangle.obj as the expected rows below imply it (resources/data is an empty submodule): the gradient rows of
# src/render/tests/test_mesh.py:568-640 name vertex 3 = (1, 1, 0) ("the 4th vertex", hit at (0.99999, 0.99999)), put
# d n.x on vertices 1 and 3 and d n.y on vertices 0 and 3 with the face normal +z, and d dp_du.x on 1 (-1) and 3 (+1):
RECT_VERTICES = np.array([[1.0, -1.0, 0.0], [-1.0, 1.0, 0.0], [-1.0, -1.0, 0.0], [1.0, 1.0, 0.0]])
RECT_TEXCOORDS = (RECT_VERTICES[:, :2] + 1.0) / 2.0           # d uv / d o.x = 0.5 (test_mesh.py:404-406)
RECT_UPPER = (0, 3, 1)                                        # the triangle under (0.99999, 0.99999); normal +z
RECT_LOWER = (0, 1, 2)                                        # the triangle under (-0.3, -0.4); normal +z

# first-vertex tangent (forward mode through Moeller-Trumbore).  src/render/tests/test_mesh.py:380-421:
# ray o = (-0.3, -0.4, -10), d = (0, 0, 1) onto the rectangle
TANGENT_RAY = ((-0.3, -0.4, -10.0), (0.0, 0.0, 1.0))
TANGENT_FORWARD = [
    # seed (which input moves, unit tangent)      expected                                      test_mesh.py
    ("o", (1, 0, 0), "p", (1.0, 0.0, 0.0)),       # :399-402
    ("o", (1, 0, 0), "uv", (0.5, 0.0)),           # :404-407
    ("o", (0, 0, 1), "t", (-1.0,)),               # :409-412
    ("d", (1, 0, 0), "p", (10.0, 0.0, 0.0)),      # :414-418
]
# reverse mode, src/render/tests/test_mesh.py:439-455: adjoint of si.p.x w.r.t. ray.o = (1,0,0); of si.t = (0,0,-1)
TANGENT_BACKWARD = [("p", 0, (1.0, 0.0, 0.0)), ("t", 0, (0.0, 0.0, -1.0))]

# gather adjoints into vertex_positions (the scatter).  src/render/tests/test_mesh.py:560-640: ray o = (0.99999, 0.99999, -10),
# d = (0, 0, 1); rows are d(seed)/d vertex_positions flattened (4 vertices x 3), atol 1e-5
SCATTER_RAY = ((0.99999, 0.99999, -10.0), (0.0, 0.0, 1.0))
SCATTER_ROWS = [
    # seeded quantity                expected gradient of vertex_positions                      test_mesh.py
    ("p.z",          [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1]),          # :576-581
    ("n.x",          [0, 0, 0, 0, 0, 0.5, 0, 0, 0, 0, 0, -0.5]),     # :611-616
    ("n.y",          [0, 0, 0.5, 0, 0, 0, 0, 0, 0, 0, 0, -0.5]),     # :618-623
    ("sh_frame.n.x", [0, 0, 0, 0, 0, 0.5, 0, 0, 0, 0, 0, -0.5]),     # :625-631
    ("sh_frame.n.y", [0, 0, 0.5, 0, 0, 0, 0, 0, 0, 0, 0, -0.5]),     # :633-639
]
SCATTER_ATOL = 1e-5

# ----------------------------------------------------------------------------------------------------------------------
# Microfacet distributions, isotropic alpha = 0.1, sample_visible = False.  src/render/tests/test_microfacet.py
MF_ALPHA = 0.1
_steps = 20


def _dirs(theta, phi):
    theta, phi = np.broadcast_arrays(np.asarray(theta, dtype=np.float64), np.asarray(phi, dtype=np.float64))
    return np.stack([np.cos(phi) * np.sin(theta), np.sin(phi) * np.sin(theta), np.cos(theta)], -1)


# directions of test02 / test03: theta = linspace(0, pi, 20), phi = pi/2 (:25-29) and theta = 0.1, phi = linspace(0, 2 pi, 20) (:69-73)
MF_DIRS_THETA_SWEEP = _dirs(np.linspace(0, math.pi, _steps), math.pi / 2)
MF_DIRS_PHI_SWEEP = _dirs(0.1, np.linspace(0, 2 * math.pi, _steps))
MF_WI = (0.0, 0.0, 1.0)
BECKMANN_EVAL_THETA_SWEEP = [3.18309879e+01, 2.07673073e+00, 3.02855828e-04, 1.01591990e-11] + [0.0] * 16          # :51-58
BECKMANN_PDF_THETA_SWEEP = [3.18309879e+01, 2.04840684e+00, 2.86446273e-04, 8.93474877e-12] + [0.0] * 16           # :60-67
BECKMANN_EVAL_PHI_SWEEP = [11.86709118] * _steps                                                                   # :87
BECKMANN_PDF_PHI_SWEEP = [11.86709118 * math.cos(0.1)] * _steps                                                    # :88
# smith_g1(v, wi): theta = linspace(pi/3, pi/2, 20), phi = pi/2 (:95-100); theta = pi/2 * 0.98, phi sweep (:121-126)
G1_DIRS_THETA_SWEEP = _dirs(np.linspace(math.pi / 3, math.pi / 2, _steps), math.pi / 2)
G1_DIRS_PHI_SWEEP = _dirs(math.pi / 2 * 0.98, np.linspace(0, 2 * math.pi, _steps))
BECKMANN_G1_THETA_SWEEP = [1.0] * 14 + [9.9828446e-01, 9.8627287e-01, 9.5088160e-01, 8.5989666e-01, 6.2535185e-01,
                                        5.7592310e-06]                                                            # :110-116 (atol 1e-5)
BECKMANN_G1_PHI_SWEEP = [0.67333597] * _steps                                                                      # :129
GGX_G1_THETA_SWEEP = [9.9261039e-01, 9.9160647e-01, 9.9042398e-01, 9.8901933e-01, 9.8733366e-01,
                      9.8528832e-01, 9.8277503e-01, 9.7964239e-01, 9.7567332e-01, 9.7054905e-01,
                      9.6378750e-01, 9.5463598e-01, 9.4187391e-01, 9.2344058e-01, 8.9569420e-01,
                      8.5189372e-01, 7.7902949e-01, 6.5144652e-01, 4.1989169e-01, 3.2584082e-06]                  # :209-215 (atol 1e-5)
GGX_G1_PHI_SWEEP = [0.46130955] * _steps                                                                           # :227

# sample(wi = (0,0,1), u): the reference's distribution there is ANISOTROPIC (alpha_u, alpha_v) = (0.1, 0.3); the rows with
# u2 in {0, 0.5} (phi = 0, pi) lie on the alpha_u axis, where the sampled normal equals the isotropic alpha = 0.1 one and the
# density D cos (both distributions carry 1 / (pi alpha_u alpha_v)) is alpha_u / alpha_v = 1/3 of the isotropic density.
# u1 = 0, 1/6, .., 5/6.  test_microfacet.py:144-188 (Beckmann), :241-285 (GGX); atol 5e-4 (normal), 1e-4 (pdf)
MF_SAMPLE_U1 = [i / 6.0 for i in range(6)]
MF_SAMPLE_ANISO_RATIO = 3.0
BECKMANN_SAMPLE_U2_0 = ([[0.0, 0.0, 1.0], [4.26597558e-02, 0.0, 9.99089658e-01], [6.35476336e-02, 0.0, 9.97978806e-01],
                         [8.29685107e-02, 0.0, 9.96552169e-01], [1.04243755e-01, 0.0, 9.94551778e-01],
                         [1.32673502e-01, 0.0, 9.91159797e-01]],
                        [10.610329, 8.866132, 7.1166167, 5.360419, 3.5952191, 1.816128])
BECKMANN_SAMPLE_U2_HALF = ([[-0.0, -0.0, 1.0], [-4.26597558e-02, -1.11883027e-08, 9.99089658e-01],
                            [-6.35476336e-02, -1.66665313e-08, 9.97978806e-01], [-8.29685107e-02, -2.17600107e-08, 9.96552169e-01],
                            [-1.04243755e-01, -2.73398335e-08, 9.94551778e-01], [-1.32673502e-01, -3.47960558e-08, 9.91159797e-01]],
                           [10.610329, 8.866132, 7.1166167, 5.360419, 3.5952191, 1.816128])
GGX_SAMPLE_U2_0 = ([[0.0, 0.0, 1.0], [4.4676583e-02, 0.0, 9.9900150e-01], [7.0534222e-02, 0.0, 9.9750936e-01],
                    [9.9504232e-02, 0.0, 9.9503714e-01], [1.4002767e-01, 0.0, 9.9014759e-01], [2.1821797e-01, 0.0, 9.7590005e-01]],
                   [10.610329, 7.390399, 4.751113, 2.6924708, 1.214469, 0.3171101])
GGX_SAMPLE_U2_HALF = ([[-0.0, -0.0, 1.0], [-4.4676583e-02, -1.1717252e-08, 9.9900150e-01], [-7.0534222e-02, -1.8498891e-08, 9.9750936e-01],
                       [-9.9504232e-02, -2.6096808e-08, 9.9503714e-01], [-1.4002767e-01, -3.6724821e-08, 9.9014759e-01],
                       [-2.1821797e-01, -5.7231659e-08, 9.7590005e-01]],
                      [10.610329, 7.390399, 4.751113, 2.6924708, 1.214469, 0.3171101])

# ----------------------------------------------------------------------------------------------------------------------
# Fresnel.  src/render/tests/test_fresnel.py:6-37: fresnel(cos_theta_i, eta) -> (F, cos_theta_t, eta_it, eta_ti)
_ct_crit = -math.sqrt(1 - 1 / 1.5 ** 2)
FRESNEL_ROWS = [
    ((1.0, 1.5), (0.04, -1.0, 1.5, 1 / 1.5)),             # :8
    ((-1.0, 1.5), (0.04, 1.0, 1 / 1.5, 1.5)),             # :9
    ((1.0, 1 / 1.5), (0.04, -1.0, 1 / 1.5, 1.5)),         # :10
    ((-1.0, 1 / 1.5), (0.04, 1.0, 1.5, 1 / 1.5)),         # :11
    ((0.0, 1.5), (1.0, _ct_crit, 1.5, 1 / 1.5)),          # :12
    ((0.0, 1 / 1.5), (1.0, 0.0, 1 / 1.5, 1.5)),           # :13
]
# spot checks against hyperphysics (:15-37): (cos_theta_i, eta) -> F, cos_theta_t (None = not stated)
FRESNEL_SPOT = [
    ((math.cos(math.radians(45)), 1.5), 0.5 * (0.09201336304552442 ** 2 + 0.3033370452904235 ** 2),
     -math.cos(math.radians(28.1255057020557))),                                                                 # :17-24
    ((math.cos(math.radians(45)), 1 / 1.5), 1.0, 0.0),                                                           # :27-29 (total internal reflection)
    ((math.cos(math.radians(10)), 1 / 1.5), 0.5 * (0.19046797197779405 ** 2 + 0.20949431963852014 ** 2),
     -math.cos(math.radians(15.098086605159006))),                                                               # :31-37
]
# :47-51 index-matched: F == 0 exactly and cos_theta_t = -cos_theta_i (atol 5e-7) for cos_theta_i = linspace(-1, 1, 20)
FRESNEL_MATCHED_COS = np.linspace(-1.0, 1.0, 20)
# :54-66 fresnel_conductor(cos, eta, k = 0) == fresnel(cos, eta) for cos = cos(linspace(0, pi/2, 20)), eta in {1.5, 1/1.5}
FRESNEL_CONDUCTOR_COS = np.cos(np.linspace(0.0, math.pi / 2, 20))
FRESNEL_CONDUCTOR_ETAS = [1.5, 1 / 1.5]
# :70-77 Snell: sin(theta_i) = 1.5 sin(theta_t), atol 1e-5, theta_i = linspace(0, pi/2, 20)
SNELL_THETA_I = np.linspace(0.0, math.pi / 2, 20)

# ----------------------------------------------------------------------------------------------------------------------
# gaussian reconstruction filter (default stddev 0.5).  src/rfilters/tests/test_rfilter.py:14-19
GAUSSIAN_ROWS = [(0.2, 0.9227, 8e-3), (2.1, 0.0, 0.0)]

# ----------------------------------------------------------------------------------------------------------------------
# perspective camera.  src/sensors/tests/test_perspective.py:6-30 (create_camera: fov 34 about x, 512 x 256 film, near 1,
# far 35, look_at(o, o + d, up = y)) and :89-135 (sample_ray_differential)
CAMERA = dict(fov=34.0, width=512, height=256, near_clip=1.0, far_clip=35.0, up=(0.0, 1.0, 0.0))
CAMERA_ORIGINS = [(1.0, 0.0, 1.5), (1.0, 4.0, 1.5)]
CAMERA_DIRECTIONS = [(0.0, 0.0, 1.0), (1.0, 0.0, 0.0)]
CAMERA_POS_SAMPLES = [(0.2, 0.6), (0.1, 0.9), (0.2, 0.2)]      # :99 pos_sample = [[0.2, 0.1, 0.2], [0.6, 0.9, 0.2]] (x row, y row)

# ----------------------------------------------------------------------------------------------------------------------
# whole BSDFs.  src/bsdfs/tests/test_dielectric.py (int_ior 1.5, ext_ior 1, specular_reflectance 0.3, specular_transmittance 0.6;
# the tracer's dielectric has no transmittance parameter -- T = 1 -- and transports radiance: the reference's radiance-mode
# weights divided by 0.6), test_diffuse.py:13-35 (default reflectance 0.5), test_twosided.py:29-45.
# rows: (wi, sample1) -> (weight, pdf, eta, wo, sampled_type): DeltaReflection = 0x20, DeltaTransmission = 0x40 (bsdf.h:31-101)
DIELECTRIC = dict(int_ior=1.5, ext_ior=1.0, reflectance=0.3, reference_transmittance=0.6)
DIELECTRIC_SAMPLE_ROWS = [
    ((0, 0, 1), 0.0, 0.3, 0.04, 1.0, (0, 0, 1), 0x20),                          # test_dielectric.py:40-46
    ((0, 0, 1), 0.05, 0.6 / 1.5 ** 2 / 0.6, 1 - 0.04, 1.5, (0, 0, -1), 0x40),   # :49-58 (radiance transport)
    ((0, 0, -1), 0.0, 0.3, 0.04, 1.0, (0, 0, -1), 0x20),                        # :71-77
    ((0, 0, -1), 0.05, 0.6 * 1.5 ** 2 / 0.6, 1 - 0.04, 1 / 1.5, (0, 0, 1), 0x40),   # :80-89
]
# spot check at 80 degrees (:141-158): reflection pdf 0.387704354691473; refraction at 41.03641052520335 degrees; and back again
DIELECTRIC_SPOT_ANGLE_DEG, DIELECTRIC_SPOT_PDF, DIELECTRIC_SPOT_REFRACTED_DEG = 80.0, 0.387704354691473, 41.03641052520335
# diffuse: pdf = cos / pi, eval = 0.5 cos / pi for wo = (sin t, 0, cos t), t = i / 19 * pi / 2, wi = (0, 0, 1) (test_diffuse.py:24-31)
DIFFUSE_DEFAULT_REFLECTANCE = 0.5
DIFFUSE_THETAS = [i / 19.0 * (math.pi / 2) for i in range(20)]
# twosided(diffuse): pdf(wi = +z, wo = +z) = 1 / pi, pdf(wi = +z, wo = -z) = 0 (test_twosided.py:29-45)
