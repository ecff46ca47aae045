#!/usr/bin/env python3
"""bench.py on libudet_exp.so with experiment knobs set first:

    make -C unsupervised_detection_amd/csrc exp
    python tools/knob_bench.py 7=3 -- --steps 30 --no-cpu-baseline ...

The knobs (ids: unsupervised_detection_amd/csrc/plan.h) do not exist in the release library -- plan_knob() is a constant 0 there.
libudet_exp.so is the same sources built with -DUDET_EXPERIMENT; this script is its only user.  It swaps the library under the package's
ctypes binding (the product loader itself has no such switch) by executing _ffi.py with the file name replaced.  A measuring aid, not
a product path: with knob 7 the results are wrong on purpose."""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load_library(lib_path):
    """The package's ctypes binding on ANOTHER build of the library (tools only; the product loader has no such switch)."""
    import torch  # noqa: F401  (before any libudet*.so: one HIP runtime per process, see _ffi.py)
    pkg = os.path.join(ROOT, "unsupervised_detection_amd")
    lib_path = os.path.abspath(lib_path)
    if not os.path.exists(lib_path):
        raise SystemExit("%s is missing (experiment build: make -C unsupervised_detection_amd/csrc exp)" % lib_path)
    import unsupervised_detection_amd  # noqa: F401  (the package itself imports nothing native)
    path = os.path.join(pkg, "_ffi.py")
    src = open(path).read().replace('os.path.join(_HERE, "libudet.so")', repr(lib_path))
    assert repr(lib_path) in src
    mod = types.ModuleType("unsupervised_detection_amd._ffi")
    mod.__file__ = path
    mod.__package__ = "unsupervised_detection_amd"
    sys.modules[mod.__name__] = mod
    exec(compile(src, path, "exec"), mod.__dict__)
    return mod.lib


def load_experiment_build():
    import ctypes
    lib = load_library(os.path.join(ROOT, "unsupervised_detection_amd", "libudet_exp.so"))
    lib.udet_exp_knob.restype = None
    lib.udet_exp_knob.argtypes = [ctypes.c_int, ctypes.c_long]
    return lib


def main():
    args = sys.argv[1:]
    cut = args.index("--") if "--" in args else len(args)
    knobs, rest = args[:cut], args[cut + 1:]
    lib = load_experiment_build()
    for k in knobs:
        i, v = k.split("=")
        lib.udet_exp_knob(int(i), int(v))
    import bench
    sys.argv = [os.path.join(ROOT, "bench.py")] + rest + ["--allow-experiment-build"]
    return bench.main()


if __name__ == "__main__":
    sys.exit(main())
