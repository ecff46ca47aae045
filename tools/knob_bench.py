#!/usr/bin/env python3
"""bench.py with experiment knobs of libudet_debug.so set first:  python tools/knob_bench.py 0=3 -- --steps 30 --no-cpu-baseline ...
(knob ids: include/udet_debug.h, udet_debug_knob).  A measuring aid, not a product path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    args = sys.argv[1:]
    cut = args.index("--") if "--" in args else len(args)
    knobs, rest = args[:cut], args[cut + 1:]
    from unsupervised_detection_amd._devel import dbg
    for k in knobs:
        i, v = k.split("=")
        dbg.udet_debug_knob(int(i), int(v))
    import bench
    sys.argv = [os.path.join(ROOT, "bench.py")] + rest
    return bench.main()


if __name__ == "__main__":
    sys.exit(main())
