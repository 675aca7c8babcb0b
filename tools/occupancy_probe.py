"""How does one fused pass scale with the number of tiles in flight?  Fixed chunk height
(CCP_GS_CHUNK) and depth (CCP_GS_TMAX), image height = chunks * CCP_GS_CHUNK.
Run under `rocprofv3 --kernel-trace` for per-dispatch durations, or read the event timing printed here.
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from coursecomputationalphotography_amd import capi  # noqa: E402


def main():
    R = int(os.environ.get("CCP_GS_CHUNK", "256"))
    T = int(os.environ.get("CCP_GS_TMAX", "8"))
    W = int(os.environ.get("PROBE_W", "16384"))
    for n in [int(a) for a in (sys.argv[1:] or "3 4 6 8 10 14 18 26 34 50 66".split())]:
        H = R * n
        g = capi.Grid(W, H, 1)
        g.randomize_x(1)
        g.b_from_x()
        g.fill_x(1.0)
        g.sweep(2 * T)
        g.synchronize()
        g.sweep(4 * T)
        g.synchronize()
        ms, launches = g.last_timing()
        per = ms / launches
        steps = R + 4 * T + 2
        print(f"W={W} H={H} chunks={n} R={R} T={T}: {per:.4f} ms/pass, {per * 1e3 / steps:.3f} us/step-if-one-round, "
              f"{W * H * T / per / 1e9:.1f} Gupd/s", flush=True)
        del g


if __name__ == "__main__":
    main()
