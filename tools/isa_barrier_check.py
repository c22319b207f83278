#!/usr/bin/env python3
"""Every s_barrier of every kernel must be reached with the wave's own LDS writes drained (s_waitcnt lgkmcnt(0) after the last ds_write / LDS atomic on EVERY path).

Why this exists (profiles/r06_seed_sort.md section 4): `__syncthreads()` is a workgroup release fence + s_barrier, and the fence's `s_waitcnt lgkmcnt(0)` is a SOFT wait the
compiler's wait-count pass may drop where its scoreboard shows no LDS operation pending.  In a diagnostic build of the seed sort it dropped the wait at the barrier that heads
the loop over a frame's global partitions -- on the path from the kernel's entry nothing is pending there, on the back edge wave 0's pushes to the segment stack are -- and
the other waves then read the stack before wave 0's writes had landed: the "failure beside a second dispatch" of rounds 4 - 6.  The library's barriers carry a hard wait
(csrc/plp_barrier.hpp); this tool reads the ISA the build keeps (csrc/build/*.s) and proves it per kernel by a forward data-flow over the basic blocks.

    python tools/isa_barrier_check.py [file.s ...]      (default: structure-plp-slam_amd/csrc/build/*.s)      exit code 1 = some barrier can be reached with LDS writes in flight
"""
import glob, os, re, sys

LABEL = re.compile(r"^(\.LBB\d+_\d+):")
FUNC = re.compile(r"^(_Z[\w$.]+):")
LDS_WRITE = re.compile(r"^\s+(ds_write|ds_wrxchg|ds_wrap|ds_add|ds_sub|ds_rsub|ds_inc|ds_dec|ds_min|ds_max|ds_and|ds_or|ds_xor|ds_mskor|ds_cmpst|ds_pk_add|ds_append|ds_ordered|ds_gws|flat_store|flat_atomic|scratch_)")
WAIT0 = re.compile(r"^\s+s_waitcnt\b.*lgkmcnt\(0\)")
BRANCH = re.compile(r"^\s+(s_branch|s_cbranch_\w+)\s+(\.LBB\d+_\d+)")


def kernels(path):
    """yield (name, [(line_no, text)]) for every function of an assembly file"""
    name, body = None, []
    for no, line in enumerate(open(path, errors="replace"), 1):
        m = FUNC.match(line)
        if m and name is None:
            name, body = m.group(1), []
            continue
        if name is not None:
            if line.startswith(".Lfunc_end"):
                yield name, body
                name = None
            else:
                body.append((no, line.rstrip("\n")))


def check(body):
    """blocks, successors, then the data-flow: state = 'an LDS write of this wave may be in flight'; returns the line numbers of barriers reached in that state"""
    blocks, cur = [], {"label": "<entry>", "ins": []}
    for no, line in body:
        m = LABEL.match(line)
        if m:
            blocks.append(cur)
            cur = {"label": m.group(1), "ins": []}
        elif line.startswith("\t") and not line.lstrip().startswith((";", ".")):
            cur["ins"].append((no, line))
    blocks.append(cur)
    index = {b["label"]: i for i, b in enumerate(blocks)}
    for i, b in enumerate(blocks):
        succ, falls = [], True
        for no, line in b["ins"]:
            m = BRANCH.match(line)
            if m:
                succ.append(index[m.group(2)])
                if m.group(1) == "s_branch":
                    falls = False
            elif re.match(r"^\s+(s_endpgm|s_setpc_b64)", line):
                falls = False
        if falls and i + 1 < len(blocks):
            succ.append(i + 1)
        b["succ"] = succ
    state_in = [False] * len(blocks)
    work, bad = list(range(len(blocks))), set()
    seen_in = [None] * len(blocks)
    while work:
        i = work.pop()
        st = state_in[i]
        if seen_in[i] == st:
            continue
        seen_in[i] = st
        for no, line in blocks[i]["ins"]:
            if WAIT0.match(line):
                st = False
            elif LDS_WRITE.match(line):
                st = True
            elif re.match(r"^\s+s_barrier", line) and st:
                bad.add(no)
        for j in blocks[i]["succ"]:
            if st and not state_in[j]:
                state_in[j] = True
                work.append(j)
            elif seen_in[j] is None:
                work.append(j)
    return sorted(bad)


def main(paths):
    rc = 0
    for path in paths:
        for name, body in kernels(path):
            n_bar = sum(1 for _, l in body if re.match(r"^\s+s_barrier", l))
            if not n_bar:
                continue
            bad = check(body)
            print(f"{os.path.basename(path)}: {name[:70]:70s} barriers {n_bar:3d}  reached with LDS writes in flight: {len(bad)}" + (f"  (lines {bad[:8]})" if bad else ""))
            rc |= bool(bad)
    return rc


if __name__ == "__main__":
    args = sys.argv[1:] or sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "structure-plp-slam_amd", "csrc", "build", "*.s")))
    sys.exit(main(args))
