"""CPU: the output path.  The library's VTU writer against the REFERENCE's own writer
(writeParticles2VTU, cuda/utils.cpp:144-283, compiled into oracle/_ref from the source where it lies), byte for
byte, and its printf-free number formatting against correctly rounded formatting on a million doubles."""
import ctypes as C
import os

import numpy as np
import pytest

from cudaparticlesfoam_amd import _lib as L


def _write(path, xyzw, cell, vel):
    lib = L.load()
    ke = C.c_double()
    r = lib.cpf_write_vtu_arrays(str(path).encode(), xyzw.shape[0], xyzw.ctypes.data, cell.ctypes.data, vel.ctypes.data,
                                 C.byref(ke))
    return r, ke.value


def _sections(text):
    """{array name: list of lines} + the skeleton (every line that is not array data)."""
    out, skel, name = {}, [], None
    for line in text.split("\n"):
        if line.startswith("<DataArray"):
            name = line.split("Name='")[1].split("'")[0]
            out[name] = []
            skel.append(line)
        elif line.startswith("</DataArray>"):
            name = None
            skel.append(line)
        elif name is not None:
            out[name].append(line)
        else:
            skel.append(line)
    return out, skel


def test_vtu_bytes_equal_the_reference_writer(tmp_path, oracle_libs):
    ref = oracle_libs.RefLib()
    rng = np.random.default_rng(3)
    n = 5000
    xyzw = np.concatenate([rng.normal(size=(n, 3)) * 10.0 ** rng.integers(-6, 3, size=(n, 1)), np.ones((n, 1))], 1)
    xyzw[::7, 3] = 0.0                                           # inactive particles (w = 0)
    xyzw[:4, :3] = [[0.0, -0.0, 1e-300], [0.5e-15, 1.5e-15, 2.5e-15], [-1.0, 123456.789, 1e12], [0.1, 0.2, 0.3]]
    cell = rng.integers(-3, 12225, size=n).astype(np.int32)
    # (a) velocities zero: the files must be identical down to the last byte
    vel0 = np.zeros((n, 4))
    mine = tmp_path / "mine.vtu"
    r, ke = _write(mine, xyzw, cell, vel0)
    assert r == 0 and ke == 0.0
    theirs = ref.write_vtu(str(tmp_path), 7, xyzw, vel0, cell, cell)
    assert open(mine, "rb").read() == open(theirs, "rb").read()
    # (b) real velocities (one NaN row -> "0 0 0" like the reference): identical except the KEs array, where the
    # reference prints 0.000000 for every NON-zero energy (utils.cpp:245-248) and this writer prints the energy
    vel = np.concatenate([rng.normal(size=(n, 3)) * 3.0, np.zeros((n, 1))], 1)
    vel[11, 0] = np.nan
    vel[12, :3] = 0.0
    r, ke = _write(mine, xyzw, cell, vel)
    theirs = ref.write_vtu(str(tmp_path), 8, xyzw, vel, cell, cell)
    a, askel = _sections(open(mine).read())
    b, bskel = _sections(open(theirs).read())
    assert askel == bskel and list(a) == list(b)
    for name in a:
        if name != "KEs":
            assert a[name] == b[name], name
    e = 0.5 * (vel[:, :3] ** 2).sum(1)
    assert a["KEs"][12] == b["KEs"][12] == "0.000000"
    assert a["KEs"][5] == "%f" % e[5] and b["KEs"][5] == "0.000000"
    assert r == L.CPF_WARN_NAN and np.isnan(ke)                  # NaN energy is reported, not "pause"d on


def test_a_large_frame_takes_the_parallel_path_and_keeps_the_bytes(tmp_path, oracle_libs):
    """150 000 particles (19 chunks per array): every array of the frame is formatted from a work counter by many threads, each
    chunk into its own string, written in order; the file must still be the reference writer's, byte for byte."""
    ref = oracle_libs.RefLib()
    rng = np.random.default_rng(17)
    n = 150_000
    xyzw = np.concatenate([rng.normal(size=(n, 3)) * 10.0 ** rng.integers(-5, 3, size=(n, 1)), np.ones((n, 1))], 1)
    xyzw[::11, 3] = 0.0
    cell = rng.integers(-2, 248000, size=n).astype(np.int32)
    vel0 = np.zeros((n, 4))
    mine = tmp_path / "big.vtu"
    r, ke = _write(mine, xyzw, cell, vel0)
    assert r == 0 and ke == 0.0
    theirs = ref.write_vtu(str(tmp_path), 3, xyzw, vel0, cell, cell)
    got = open(mine, "rb").read()
    assert len(got) > 12_000_000 and got == open(theirs, "rb").read()


def test_number_formatting_is_correctly_rounded(tmp_path):
    """'%.15lf' / '%lf' without printf: exact decimal expansion, round-half-even at the last digit -- compared with
    Python's correctly rounded formatting over every binade, ties, zeros, denormals, huge and non-finite values."""
    rng = np.random.default_rng(1)
    m = 133_000
    vals = np.concatenate([
        rng.random(m), rng.normal(size=m) * 10.0 ** rng.integers(-20, 15, m),
        np.ldexp(rng.random(m), rng.integers(-1074, 62, m)) * rng.choice([-1.0, 1.0], m),
        [0.0, -0.0, 0.5e-15, 1.5e-15, 2.5e-15, 0.5e-6, 1.5e-6, 2.5e-6, 1e-300, 5e-324, -5e-324, 1.0, -1.0,
         123456789012345.0, 9.2e18, -9.2e18, 1e19, 1e300, np.inf, -np.inf, 0.1, 0.2, 0.3, 1e15 + 0.5, 2.0 ** 53, 2.0 ** 62,
         0.000000000000000499999, 0.9999999999999995, 0.99999949999, 18446.744073709552, 18446.744073709553]])
    vals = vals[: (vals.size // 3) * 3]
    n = vals.size // 3
    xyzw = np.concatenate([vals.reshape(n, 3), np.ones((n, 1))], 1)
    vel = np.concatenate([vals.reshape(n, 3), np.zeros((n, 1))], 1)
    vel[np.isnan(vel[:, 0]), 0] = 1.0
    cell = np.arange(n, dtype=np.int32) - 5
    path = tmp_path / "fmt.vtu"
    r, _ = _write(path, xyzw, cell, vel)
    assert r in (L.CPF_OK, L.CPF_WARN_NAN)
    sec, _ = _sections(open(path).read())
    want15 = ["%.15f %.15f %.15f" % tuple(row) for row in xyzw[:, :3]]
    assert sec["Position"] == want15
    want6 = ["%f %f %f" % tuple(row) for row in vel[:, :3]]
    assert sec["vels"] == want6
    assert sec["ParticleTetID"] == [str(int(c)) for c in cell] and sec["offsets"] == [str(i + 1) for i in range(n)]


def test_unwritable_path_is_an_error(tmp_path):
    x = np.zeros((1, 4)); c = np.zeros(1, np.int32)
    r, _ = _write(tmp_path / "no_such_dir" / "p.vtu", x, c, x)
    assert r == L.CPF_ERR_ARG


def _read_appended(path):
    """Minimal reader of a VTU with raw appended data (header_type UInt64): {array name: numpy array}."""
    raw = open(path, "rb").read()
    k = raw.index(b"<AppendedData encoding='raw'>")
    head = raw[:k].decode()
    start = raw.index(b"_", k) + 1
    np_type = {"Float64": np.float64, "Float32": np.float32, "Int32": np.int32, "UInt8": np.uint8}
    out = {}
    for line in head.split("\n"):
        if not line.startswith("<DataArray"):
            continue
        a = {kv.split("=")[0]: kv.split("=")[1].strip("'/>") for kv in line[len("<DataArray "):].split(" ")}
        assert a["format"] == "appended"
        off = start + int(a["offset"])
        nbytes = int(np.frombuffer(raw[off:off + 8], np.uint64)[0])
        arr = np.frombuffer(raw[off + 8:off + 8 + nbytes], np_type[a["type"]])
        comps = int(a["NumberOfComponents"])
        out[a["Name"]] = arr.reshape(-1, comps) if comps > 1 else arr
    assert raw.rstrip().endswith(b"</VTKFile>") and b"NumberOfCells='%d' NumberOfPoints='%d'" % (len(out["types"]), len(out["types"])) in raw
    return out


def test_binary_appended_frame_holds_the_ascii_frames_data(tmp_path):
    """"binary-appended as an option".  cpf_write_vtu_arrays_binary writes the same DataArrays -- names, types,
    components, order -- with format='appended' and raw little-endian blocks; read back, they are the inputs exactly (positions
    to the last bit, where the ASCII frame has 15 decimals), and what the ASCII frame of the same particles says."""
    lib = L.load()
    rng = np.random.default_rng(8)
    n = 3001
    xyzw = np.concatenate([rng.normal(size=(n, 3)) * 10.0 ** rng.integers(-6, 3, size=(n, 1)), np.ones((n, 1))], 1)
    xyzw[::5, 3] = 0.0
    cell = rng.integers(-3, 12225, size=n).astype(np.int32)
    vel = np.concatenate([rng.normal(size=(n, 3)) * 3.0, np.zeros((n, 1))], 1)
    vel[11, 0] = np.nan                                           # "vels" prints / stores 0 0 0 there, the energy is NaN
    ke = C.c_double()
    b = tmp_path / "frame_bin.vtu"
    r = lib.cpf_write_vtu_arrays_binary(str(b).encode(), n, xyzw.ctypes.data, cell.ctypes.data, vel.ctypes.data, C.byref(ke))
    assert r == L.CPF_WARN_NAN and np.isnan(ke.value)            # like the ASCII writer: NaN energy is reported, the file is written
    vel[11, 0] = 0.25
    r = lib.cpf_write_vtu_arrays_binary(str(b).encode(), n, xyzw.ctypes.data, cell.ctypes.data, vel.ctypes.data, C.byref(ke))
    assert r == 0
    d = _read_appended(b)
    assert list(d) == ["Position", "ParticleType", "ParticleID", "ParticleTetID", "ConvexTetID", "vels", "KEs", "connectivity",
                       "offsets", "types"]
    assert np.array_equal(d["Position"], xyzw[:, :3]) and np.array_equal(d["ParticleType"], xyzw[:, 3].astype(np.int32))
    assert np.array_equal(d["ParticleID"], np.arange(n)) and np.array_equal(d["ParticleTetID"], cell) and np.array_equal(d["ConvexTetID"], cell)
    assert np.array_equal(d["vels"], vel[:, :3].astype(np.float32))
    e = 0.5 * (vel[:, :3] ** 2).sum(1)
    assert np.array_equal(d["KEs"], e.astype(np.float32)) and abs(ke.value - e.sum()) < 1e-9 * e.sum()
    assert np.array_equal(d["connectivity"], np.arange(n)) and np.array_equal(d["offsets"], np.arange(1, n + 1)) and (d["types"] == 1).all()
    # against the ASCII frame of the same particles
    a = tmp_path / "frame_ascii.vtu"
    r, ke_a = _write(a, xyzw, cell, vel)
    assert r == 0 and ke_a == ke.value
    sec, _ = _sections(open(a).read())
    pos_a = np.array([[float(v) for v in ln.split()] for ln in sec["Position"] if ln])
    assert (np.abs(pos_a - d["Position"]) <= 5.01e-16 + 2.3e-16 * np.abs(d["Position"])).all()     # 15 decimals + the parse's rounding
    assert [int(v) for v in sec["ConvexTetID"] if v] == list(d["ConvexTetID"])
    assert os.path.getsize(b) < os.path.getsize(a)
    # empty cloud: a valid file with empty blocks
    r = lib.cpf_write_vtu_arrays_binary(str(b).encode(), 0, None, None, None, C.byref(ke))
    assert r == 0 and len(_read_appended(b)["types"]) == 0
