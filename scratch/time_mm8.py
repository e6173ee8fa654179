"""Timing-only driver for the int8 matrix-core mat-vec at the config-3 encode shape (for rocprofv3)."""
import sys

sys.path.insert(0, "scratch")
import check_mm8 as T  # noqa: E402

ctx = T.Context.get(T.P)
T.timing(ctx, int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 22,
         reps=int(sys.argv[3]) if len(sys.argv) > 3 else 10, layout=sys.argv[4] if len(sys.argv) > 4 else "col")
