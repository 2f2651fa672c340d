"""CPU: Philox4x32-10 known-answer vectors (Random123 kat_vectors) and the normal transform."""
import numpy as np


def test_philox_known_answers(oracle_libs):
    cw = oracle_libs.CellWalk()
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, want in kat:
        got = cw.philox(np.array(ctr, np.uint32), np.array(key, np.uint32))
        assert tuple(int(v) for v in got) == want


def test_seven_round_variant_is_the_same_function_three_rounds_earlier(oracle_libs):
    """The kernels draw from Philox4x32-7 (csrc/cpf_walk.h: the fewest rounds reported as passing BigCrush).  Random123's
    published vectors are for R = 10, checked above; R = 7 runs the same round function and key schedule: three more rounds
    applied to its output, with the key where the schedule has it after seven bumps, give the R = 10 vector."""
    cw = oracle_libs.CellWalk()
    ctr, key = (0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)
    w7 = cw.philox(np.array(ctr, np.uint32), np.array(key, np.uint32), rounds=7)
    k7 = np.array([(key[0] + 7 * 0x9E3779B9) & 0xFFFFFFFF, (key[1] + 7 * 0xBB67AE85) & 0xFFFFFFFF], np.uint32)
    w10 = cw.philox(w7, k7, rounds=3)
    assert tuple(int(v) for v in w10) == (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)
    assert not np.array_equal(w7, w10)
    # and the words the deviates are made from are the seven-round ones
    n3 = cw.normal3(12345, 6, 77)
    w = cw.philox(np.array([12345, 0, 6, 0], np.uint32), np.array([77, 0x43504631], np.uint32), rounds=7)
    assert np.array_equal(n3, cw.normal3_words(w))
    # avalanche at R = 7: flipping one counter bit flips about half of the 128 output bits
    flips = []
    for bit in range(0, 64, 3):
        c2 = np.array([12345 ^ (1 << bit) if bit < 32 else 12345, (1 << (bit - 32)) if bit >= 32 else 0, 6, 0], np.uint32)
        w2 = cw.philox(c2, np.array([77, 0x43504631], np.uint32), rounds=7)
        flips.append(sum(bin(int(a) ^ int(b)).count("1") for a, b in zip(w, w2)))
    assert 52 < np.mean(flips) < 76 and min(flips) > 36


def test_normal3_moments(oracle_libs):
    cw = oracle_libs.CellWalk()
    xi = np.array([cw.normal3(g, 3, 42) for g in range(20000)])
    assert np.abs(xi.mean(0)).max() < 0.03 and np.abs(xi.var(0) - 1).max() < 0.04
    assert abs(np.corrcoef(xi.T)[0, 1]) < 0.03
    assert not np.array_equal(cw.normal3(5, 3, 42), cw.normal3(5, 4, 42))     # step enters the counter
    assert np.array_equal(cw.normal3(5, 3, 42), cw.normal3(5, 3, 42))         # stateless / reproducible


def test_normal_transform_known_answers_and_reach(oracle_libs):
    """The deviate transform on given Philox words (cw_normal3_words = the kernels' normal3 with libm): the radius
    uniform uses all 32 bits, u = (w + 0.5) * 2^-32, so the smallest word gives sqrt(2 * 33 * ln 2) = 6.7637 sigma
    (23-bit uniforms stopped at 5.77) and the largest gives 0, never NaN; pinned values for four word sets."""
    cw = oracle_libs.CellWalk()
    top = cw.normal3_words([0, 0, 0, 0])
    assert abs(np.hypot(top[0], top[1]) - np.sqrt(2 * 33 * np.log(2))) < 1e-5 and abs(top[2]) > 6.6
    assert np.all(np.isfinite(cw.normal3_words([0xFFFFFFFF] * 4))) and np.abs(cw.normal3_words([0xFFFFFFFF] * 4)).max() < 1e-3
    # the radius is a decreasing function of its word across float rounding and exponent boundaries
    words = [0, 1, 2, 255, 256, (1 << 23) - 1, 1 << 23, (1 << 24) + 1, (1 << 31) - 1, 1 << 31, 0xFFFFFF00, 0xFFFFFFFF]
    radii = [float(np.hypot(*cw.normal3_words([w, 12345 << 9, 0, 0])[:2])) for w in words]
    assert all(a >= b for a, b in zip(radii, radii[1:])) and radii[0] > 6.76 and radii[-1] < 1e-3
    for w, want_r in ((1, 6.59928), (1 << 16, 4.70964), (1 << 31, 1.17741), (3 << 30, 0.75853)):
        got = cw.normal3_words([w, 0, w, 0])
        u = (w + 0.5) / 2.0 ** 32
        assert abs(np.hypot(got[0], got[1]) - np.sqrt(-2 * np.log(u))) < 2e-6 * max(1.0, np.sqrt(-2 * np.log(u)))
        assert abs(np.hypot(got[0], got[1]) - want_r) < 1e-4
    kat = {(0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8): None}          # Philox KAT output 1 as transform input
    for wds in kat:
        a = cw.normal3_words(list(wds))
        u0, u1 = (wds[0] + 0.5) / 2.0 ** 32, ((wds[1] >> 9) + 0.5) / 2.0 ** 23
        u2, u3 = (wds[2] + 0.5) / 2.0 ** 32, ((wds[3] >> 9) + 0.5) / 2.0 ** 23
        ref = np.array([np.sqrt(-2 * np.log(u0)) * np.cos(2 * np.pi * u1), np.sqrt(-2 * np.log(u0)) * np.sin(2 * np.pi * u1),
                        np.sqrt(-2 * np.log(u2)) * np.cos(2 * np.pi * u3)])
        assert np.abs(a - ref).max() < 5e-6


def test_brownian_cellwalk_variance(oracle_libs):
    from cudaparticlesfoam_amd.cases import box_mesh
    cw = oracle_libs.CellWalk()
    mesh = box_mesh(2, 2, 2, lower=(-1, -1, -1), upper=(1, 1, 1))
    t = cw.build(mesh)
    n = 50000
    x = np.full(n, 1e-3); y = x.copy(); z = x.copy(); c = np.full(n, 7, np.int32)
    D, dt = 1.5e-5, 1e-4
    cw.step(x, y, z, c, dt, 1, t, np.zeros((8, 3)), nthreads=cw.max_threads, D=D, seed=9)
    s2 = 2 * D * dt
    for a in (x, y, z):
        assert abs((a - 1e-3).var() / s2 - 1) < 0.03
