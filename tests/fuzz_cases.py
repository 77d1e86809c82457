"""Re-run named cases of tests/fuzz_grad.py (the `BAD` lines of its log) with whatever library SVGP_MI355X_LIB selects, and
print every block's error against the oracle: python tests/fuzz_cases.py <fuzz log> [more logs]"""
import ast
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_grad as fz  # noqa: E402


def main():
    cases = []
    for path in sys.argv[1:]:
        for line in open(path):
            if line.strip().startswith("BAD"):
                tag = ast.literal_eval(line.strip()[3:].strip())[0]
                tag["dtype"] = getattr(np, tag["dtype"])
                cases.append(tag)
    ctx = fz._ffi.Context(0)
    for c in cases:
        errs = fz.run_case(ctx, c)
        print({k: (v.__name__ if k == "dtype" else v) for k, v in c.items() if k in ("N", "M", "d", "family", "lik", "dtype", "centered")},
              {k: f"{v:.1e}" for k, v in errs.items()}, flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
