"""CPU: libudet.so loads, exports every symbol include/udet.h declares, and its parameter tables agree with the
oracle's independent restatement of the reference's variable lists."""
import ctypes
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from unsupervised_detection_amd import _ffi
    def declared(header):
        hdr = open(os.path.join(ROOT, "include", header)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        return set(re.findall(r"\b(udet_[a-z0-9_]+)\s*\(", hdr))
    names = declared("udet.h")
    assert len(names) >= 25
    for n in sorted(names):
        assert hasattr(_ffi.lib, n), f"libudet.so does not export {n}"
    assert _ffi.lib.udet_version() >= 100
    # the test-only hooks are NOT in the product library: they live in libudet_debug.so (include/udet_debug.h)
    from unsupervised_detection_amd import _devel
    hooks = declared("udet_debug.h")
    assert len(hooks) >= 4
    for n in sorted(hooks):
        assert hasattr(_devel.dbg, n), f"libudet_debug.so does not export {n}"
        assert not hasattr(_ffi.lib, n), f"libudet.so must not export the test hook {n}"


def test_release_library_has_no_experiment_knobs():
    """the work-skipping ablation mask (UDET_KNOB_SKIP) and the lane knobs must not be reachable in libudet.so.  They are
    compiled out (csrc/plan.h: plan_knob() is a constant 0 without -DUDET_EXPERIMENT): no knob table, no setter, mangled or not."""
    import subprocess
    from unsupervised_detection_amd import _ffi
    assert not hasattr(_ffi.lib, "udet_exp_knob")
    nm = None
    for tool in ("nm", "/opt/rocm/lib/llvm/bin/llvm-nm"):
        try:
            nm = subprocess.run([tool, "-D", "--defined-only", _ffi.LIB_PATH], capture_output=True, text=True, check=True).stdout
            break
        except (OSError, subprocess.CalledProcessError):
            continue
    assert nm is not None, "no nm tool"
    assert "knob" not in nm.lower(), [l for l in nm.splitlines() if "knob" in l.lower()]


def test_no_kernel_keeps_locals_in_scratch_memory(tmp_path):
    """Round 6: every LDS-DMA kernel carried 12 bytes of private segment since round 5 -- two captured counters behind a pointer phi -- and
    with them a scratch load + `s_waitcnt vmcnt(0)` in front of every stage issue (profiles/NOTES.md).  Guard: a kernel of libudet.so may
    use scratch memory only because it spills registers under a launch-bounds cap (the 2-stage 128 x 128 / 128 x 96 tiles, the 4-wave
    Winograd kernel: vgpr_spill_count > 0), never for a local variable.  Reads the gfx950 code objects out of the library's offload
    bundles (what tools/scratch_report.sh reports per source file)."""
    import re
    import struct
    import subprocess
    from unsupervised_detection_amd import _ffi
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.exists(readelf):
        pytest.skip("no llvm-readelf")
    blob = open(_ffi.LIB_PATH, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    kernels, pos, n_objects = {}, 0, 0
    while True:
        i = blob.find(magic, pos)
        if i < 0:
            break
        pos = i + len(magic)
        (count,) = struct.unpack_from("<Q", blob, i + 24)
        o = i + 32
        for _ in range(count):
            off, size, tlen = struct.unpack_from("<QQQ", blob, o)
            o += 24
            triple = blob[o:o + tlen].decode()
            o += tlen
            if "gfx950" not in triple or size == 0:
                continue
            co = tmp_path / f"co_{n_objects}.co"
            co.write_bytes(blob[i + off:i + off + size])
            n_objects += 1
            notes = subprocess.run([readelf, "--notes", str(co)], capture_output=True, text=True, check=True).stdout
            name = None
            for line in notes.splitlines():
                m = re.match(r"\s+\.(name|private_segment_fixed_size|vgpr_spill_count):\s+(\S+)", line)
                if not m:
                    continue
                if m.group(1) == "name":
                    name = m.group(2)
                    kernels.setdefault(name, {})
                elif name is not None:
                    kernels[name][m.group(1)] = int(m.group(2))
    assert n_objects > 0 and len(kernels) > 100, (n_objects, len(kernels))
    bad = sorted(k for k, v in kernels.items() if v.get("private_segment_fixed_size", 0) > 0 and v.get("vgpr_spill_count", 0) == 0)
    assert not bad, bad


def test_tune_cache_file_round_trip(tmp_path):
    """udet_tune_save / udet_tune_load (host only): the text form of the autotuner's choices.  A line carries
    `c <key> bm bn ks ws fold tail`; the header names the build's tuning ABI and a file of another build (or anything else) is
    rejected (a stale file must not feed configurations to a different kernel set)."""
    from unsupervised_detection_amd import _ffi
    lib = _ffi.lib
    lib.udet_tune_load.restype = ctypes.c_int
    lib.udet_tune_save.restype = ctypes.c_int
    lib.udet_tuned_shapes.restype = ctypes.c_int
    f = tmp_path / "tune.txt"
    g0 = tmp_path / "hdr.txt"
    assert lib.udet_tune_save(str(g0).encode()) == 0
    header = g0.read_text().splitlines()[0]
    assert header.startswith("udet-tune 2 abi ")
    f.write_text(header + "\nc 1111 64 64 4 2 0 256\nc 2222 128 128 1 4 0\nw 3333 1048604\nnot a line\n")
    before = lib.udet_tuned_shapes()
    assert lib.udet_tune_load(str(f).encode()) == 3
    assert lib.udet_tuned_shapes() == before + 3
    g = tmp_path / "out.txt"
    assert lib.udet_tune_save(str(g).encode()) == 0
    lines = g.read_text().splitlines()
    assert lines[0] == header
    assert "c 1111 64 64 4 2 0 256" in lines and "c 2222 128 128 1 4 0 0" in lines and "w 3333 1048604" in lines
    bad = tmp_path / "bad.txt"
    bad.write_text("something else\n")
    assert lib.udet_tune_load(str(bad).encode()) < 0
    old = tmp_path / "old.txt"
    old.write_text("udet-tune 1\nc 1111 64 64 4 2 0 256\n")  # a file of an earlier build
    assert lib.udet_tune_load(str(old).encode()) < 0
    other = tmp_path / "other.txt"
    other.write_text("udet-tune 2 abi 999999\nc 1111 64 64 4 2 0 256\n")
    assert lib.udet_tune_load(str(other).encode()) < 0
    assert lib.udet_tune_load(str(tmp_path / "missing.txt").encode()) < 0


def test_param_tables_match_oracle():
    from oracle import oracle_torch as O
    from unsupervised_detection_amd import weights as W
    for net, specs in ((W.NET_PWC, O.pwc_param_specs()), (W.NET_GEN, O.generator_param_specs()), (W.NET_REC, O.recover_param_specs())):
        tab = W.param_table(net)
        assert [(n, tuple(s)) for n, s, _ in tab] == [(n, tuple(s)) for n, s, _ in specs]
        off = 0
        for n, s, o in tab:
            assert o == off
            off += int(np.prod(s))
        assert off == W.param_total(net)
    assert W.param_total(W.NET_PWC) == 14079050 and W.param_total(W.NET_GEN) == 1451062 and W.param_total(W.NET_REC) == 3388610


def test_init_families():
    from unsupervised_detection_amd import weights as W
    g = W.as_dict(W.init_flat(W.NET_GEN), W.NET_GEN)
    assert float(g["MaskNet/conv1/bn/gamma"].min()) == 1.0 and float(g["MaskNet/conv1/bias"].abs().max()) == 0.0
    k = g["MaskNet/conv5/kernel"]
    lim = (6.0 / (9 * 128 + 9 * 128)) ** 0.5
    assert float(k.abs().max()) <= lim and float(k.abs().max()) > 0.9 * lim
    p = W.as_dict(W.init_flat(W.NET_PWC), W.NET_PWC)
    k = p["pwcnet/featpyr/conv3aa/kernel"]
    std = (2.0 / (9 * 64)) ** 0.5 / 0.87962566103423978
    assert float(k.abs().max()) <= 2 * std + 1e-6 and abs(float(k.std()) - (2.0 / (9 * 64)) ** 0.5) < 0.1 * std


def test_plan_rejects_bad_config_without_gpu():
    from unsupervised_detection_amd import _ffi, engine
    c = engine._Cfg(4, 100, 640, 192, 384, 80, .5, 75, 1e-4, .9, .999, 1e-8, .2, 1)
    h = ctypes.c_void_p()
    assert _ffi.lib.udet_plan_create(ctypes.byref(c), ctypes.byref(h)) != 0
    assert b"multiples of 64" in _ffi.lib.udet_last_error()
    c = engine._Cfg(4, 384, 640, 192, 384, 80, .5, 75, 1e-4, .9, .999, 1e-8, .2, 1)
    assert _ffi.lib.udet_plan_create(ctypes.byref(c), ctypes.byref(h)) == 0
    gb = _ffi.lib.udet_workspace_bytes(h) / 2 ** 30
    assert 0.5 < gb < 16, gb
    _ffi.lib.udet_plan_destroy(h)


def test_tf_checkpoint_names_map_onto_the_parameter_tables():
    """tests/golden/names.json lists the variables the reference's own code creates (oracle/make_golden.py): every one
    of them canonicalises to exactly one entry of the library's parameter tables, shapes included."""
    import json
    from unsupervised_detection_amd import weights as W
    with open(os.path.join(ROOT, "tests", "golden", "names.json")) as f:
        created = json.load(f)["variables_created_by_the_reference"]
    tables = {}
    for net in (W.NET_PWC, W.NET_GEN, W.NET_REC):
        tables.update({n: tuple(s) for n, s, _ in W.param_table(net)})
    seen = {}
    for group in created.values():
        for v in group:
            c = W.canonical_name(v["tf"] + ":0")
            if v["tf"].endswith(("moving_mean", "moving_variance")):
                assert c is None
                continue
            assert c == v["canonical"] and tables[c] == tuple(v["shape"]), v
            seen[c] = True
    assert set(seen) == set(tables)
    assert W.canonical_name("MaskNet//conv1/kernel/Adam_1") is None and W.canonical_name("train_op/global_step") is None
    # a {tf name: array} export loads directly
    import numpy as np
    export = {v["tf"]: np.full(v["shape"], 0.5, np.float32) for v in created["generator_net + recover_net"]}
    flat = W.from_tf_dict(export, W.NET_GEN)
    assert flat.numel() == W.param_total(W.NET_GEN) and float(flat.min()) == 0.5
