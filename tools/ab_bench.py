#!/usr/bin/env python3
"""bench.py on another build of the library, for same-box A/B comparisons (boxes differ by up to 10 %):

    python tools/ab_bench.py --lib gpurun_exp/libudet_base.so -- --steps 40 --no-cpu-baseline --cycles 0 --ensemble-frames 0

A measuring aid: the product loader (unsupervised_detection_amd/_ffi.py) always loads the in-tree libudet.so."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import knob_bench  # noqa: E402


def main():
    args = sys.argv[1:]
    cut = args.index("--") if "--" in args else len(args)
    mine, rest = args[:cut], args[cut + 1:]
    lib = mine[mine.index("--lib") + 1]
    knob_bench.load_library(lib)
    import bench
    sys.argv = [os.path.join(knob_bench.ROOT, "bench.py")] + rest + ["--allow-experiment-build"]
    return bench.main()


if __name__ == "__main__":
    sys.exit(main())
