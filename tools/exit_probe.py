"""Root-causing the at-exit abort recorded in round 2 (gpurun_out/exit_v4.txt: "double free or corruption (!prev)"
when torch is imported AFTER libccp_gs.so has used RCCL).  Each variant runs in a child process and prints which
libamdhip64 / librccl / libhsa-runtime64 copies are mapped before it exits; a variant that dies is run once more under
rocgdb (batch mode) for the C stack of the abort.  (An in-process SIGABRT handler printing a backtrace hangs here: the
abort comes from inside free() during exit handlers.)

  python tools/exit_probe.py            -> runs every variant as a child, one report each
  python tools/exit_probe.py <variant>  -> the child itself
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def mapped():
    seen = {}
    with open("/proc/self/maps") as fh:
        for line in fh:
            p = line.split()
            if len(p) >= 6 and any(k in p[5] for k in ("libamdhip64", "librccl", "libhsa-runtime64", "libtorch_hip", "libc10_hip", "libccp_gs", "librocm_smi")):
                seen[p[5]] = True
    return sorted(seen)


def torch_lib(name):
    import importlib.util
    spec = importlib.util.find_spec("torch")
    return os.path.join(os.path.dirname(spec.origin), "lib", name)


def child(variant):
    if variant in ("rccl_by_path_then_torch", "rccl_by_path_no_torch"):
        os.environ["CCP_GS_RCCL_LIB"] = torch_lib("librccl.so")       # torch's RCCL bound by hand, torch not imported yet
    if variant == "rocm_runtime_then_torch":
        os.environ["CCP_GS_NO_TORCH_HIP"] = "1"                        # /opt/rocm's HIP + RCCL first, torch afterwards
    from coursecomputationalphotography_amd import capi
    print(variant, "torch imported before the communicator:", "torch" in sys.modules, flush=True)
    st = None
    try:
        comm = capi.Comm(capi.comm_unique_id(), 0, 1, 0)
        print(variant, comm.info(), flush=True)
        comm.close()
    except capi.CcpError as e:
        st = e.status
        print(variant, "communicator refused:", e, flush=True)
    print(variant, "torch imported after the communicator call:", "torch" in sys.modules, flush=True)
    if variant.endswith("then_torch"):
        import torch
        print(variant, "torch sees a GPU:", torch.cuda.is_available(), flush=True)
    for p in mapped():
        print(variant, "mapped:", p, flush=True)
    print(variant, "end (status %s); interpreter exit follows" % st, flush=True)


def main():
    if len(sys.argv) > 1:
        return child(sys.argv[1])
    env = dict(os.environ)
    env["CCP_GS_DEBUG"] = "1"
    only = os.environ.get("EXIT_PROBE_VARIANTS")
    names = only.split(",") if only else ["default_order", "rccl_by_path_no_torch", "rccl_by_path_then_torch", "rocm_runtime_then_torch"]
    for v in names:
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), v], env=env, capture_output=True, text=True, timeout=150)
            rc, out, err = r.returncode, r.stdout, r.stderr
        except subprocess.TimeoutExpired as e:
            rc, out, err = "timeout", (e.stdout or b"").decode(errors="replace"), (e.stderr or b"").decode(errors="replace")
        print(f"===== {v}: exit code {rc}", flush=True)
        print(out[-6000:], flush=True)
        print(err[-3000:], flush=True)
        if rc not in (0,):
            gdb = "/opt/rocm/bin/rocgdb"
            if os.path.exists(gdb):
                cmd = ["timeout", "-k", "10", "240", gdb, "-batch", "-ex", "set pagination off", "-ex", "run", "-ex", "bt 40",
                       "-ex", "info sharedlibrary", "--args", sys.executable, os.path.abspath(__file__), v]
                try:
                    g = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
                    print(f"===== {v} under rocgdb: exit code {g.returncode}", flush=True)
                    print(g.stdout[-12000:], flush=True)
                    print(g.stderr[-3000:], flush=True)
                except subprocess.TimeoutExpired:
                    print(f"===== {v} under rocgdb: timed out", flush=True)


if __name__ == "__main__":
    main()
