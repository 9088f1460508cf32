"""Builds de6d_amd/csrc/libdet6d_hip.so (gfx950 only) with hipcc.  Works without a GPU
(hipcc cross-compiles); the built library is git-ignored but travels with gpurun snapshots."""
import json
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libdet6d_hip.so")
SOURCES = ["runtime.hip", "fps.hip", "fps_cells.hip", "fps_seq.hip", "fps_coop.hip", "ball_query.hip", "ball_query_grid.hip", "points.hip", "iou3d_nms.hip", "iou3d_host.hip", "linear.hip", "mlp_chain.hip", "mlp_group.hip", "mlp_rows.hip", "compact.hip", "expand.hip", "prepare.hip", "annos.hip", "slope.hip", "kitti_eval.hip"]
#: kernels that exist only in the -DDET6D_EXPERIMENTS library (measured alternatives that did not earn their place)
EXPERIMENT_SOURCES = []
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "fps_multi.h"), os.path.join(CSRC, "compact_parts.h"),
           os.path.join(HERE, "..", "include", "det6d_ops.h"),
           os.path.join(HERE, "..", "include", "det6d_math.h"),
           os.path.join(HERE, "..", "include", "det6d_geom.h"),
           os.path.join(HERE, "..", "include", "det6d_rng.h"),
           os.path.join(HERE, "..", "include", "det6d_riou.h")]
# -ffp-contract=off + correctly rounded divide: the arithmetic contract shared with the oracle
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt",
         # MFMA accumulators in architectural VGPRs: no v_accvgpr_read before the epilogue / the next layer
         # (vector-ALU instructions are matrix time on gfx950 for fp32 MFMA)
         "-mllvm", "-amdgpu-mfma-vgpr-form", "-fvisibility=hidden", "-Wno-unused-value"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, experiments=False, knobs=False):
    """experiments=True: a SEPARATE library (libdet6d_hip_experiments.so) compiled with -DDET6D_EXPERIMENTS, in which the
    tile-sweep / timing / stand-in environment variables of scripts/experiments are live.  The shipped library ignores them."""
    # knobs=True: a third library (libdet6d_hip_knobs.so, -DDET6D_KNOBS): the shipped kernels with the route / tile switches
    # live and no instrumentation — the flavour A/B runs of routes are taken in (csrc/common.h)
    objdir = os.path.join(CSRC, "build_experiments" if experiments else "build_knobs" if knobs else "build")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    flags = FLAGS + (["-DDET6D_EXPERIMENTS"] if experiments else ["-DDET6D_KNOBS"] if knobs else [])
    lib = LIB.replace(".so", "_experiments.so") if experiments else LIB.replace(".so", "_knobs.so") if knobs else LIB

    sources = SOURCES + (EXPERIMENT_SOURCES if experiments else [])

    def compile_one(src):
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + HEADERS):
            # the compiler's per-kernel resource report goes to <object>.usage.json (tests/test_build_resources.py: the
            # latency-chain kernels must not touch scratch memory — round 4 lost 20 % of a sampler to an array the compiler
            # had quietly moved there)
            cmd = [hipcc] + flags + ["-Rpass-analysis=kernel-resource-usage", "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            proc = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
            usage, rest = _parse_usage(proc.stderr)
            if rest.strip():
                sys.stderr.write(rest)
            if proc.returncode != 0:
                raise subprocess.CalledProcessError(proc.returncode, cmd)
            with open(o + ".usage.json", "w") as f:
                json.dump(usage, f)
        return o

    with ThreadPoolExecutor(max_workers=min(6, len(sources))) as ex:
        objs = list(ex.map(compile_one, sources))
    if force or _stale(lib, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    merged = {}
    for o in objs:
        if os.path.exists(o + ".usage.json"):
            with open(o + ".usage.json") as f:
                merged[os.path.basename(o).replace(".o", ".hip")] = json.load(f)
    with open(lib.replace(".so", ".usage.json"), "w") as f:
        json.dump(merged, f, indent=1, sort_keys=True)
    return lib


_REMARK = re.compile(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|"
                     r"SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]|Dynamic Stack):\s+(\S+)")


def _parse_usage(stderr):
    """(-Rpass-analysis=kernel-resource-usage) -> {mangled kernel name: {field: value}}, and the rest of stderr"""
    usage, rest, cur = {}, [], None
    lines = stderr.splitlines(True)
    skip = 0
    for i, line in enumerate(lines):
        if skip:
            skip -= 1
            continue
        m = _REMARK.search(line)
        if m:
            key, val = m.group(1), m.group(2)
            if key == "Function Name":
                cur = usage.setdefault(val, {})
                # clang echoes the source line and a caret under the first remark of a kernel
                j = i + 1
                while j < len(lines) and j <= i + 2 and (lines[j].lstrip().startswith("|") or re.match(r"\s*\d+ \|", lines[j])):
                    j += 1
                skip = j - i - 1
            elif cur is not None:
                cur[key.split(" [")[0]] = int(val) if val.lstrip("-").isdigit() else val
            continue
        if "[-Rpass-analysis=kernel-resource-usage]" in line:
            continue
        rest.append(line)
    return usage, "".join(rest)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, experiments="--experiments" in sys.argv, knobs="--knobs" in sys.argv))
