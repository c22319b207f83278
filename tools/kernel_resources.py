#!/usr/bin/env python3
"""Per-kernel resources of the current build (csrc/build/*.remarks): VGPRs, spilled SGPRs, scratch, occupancy, static LDS.  python tools/kernel_resources.py [substring]"""
import glob, os, re, sys
B = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "structure-plp-slam_amd", "csrc", "build")
sub = sys.argv[1] if len(sys.argv) > 1 else ""
for path in sorted(glob.glob(os.path.join(B, "*.remarks"))):
    name, d = None, {}
    for line in open(path, errors="replace"):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1); d[name] = {}; continue
        m = re.search(r"remark:(?: [^:\s]+:\d+:\d+:)?\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and name:
            d[name][m.group(1).strip()] = int(m.group(2))
    for k, v in d.items():
        if sub in k:
            print(f"{k[:70]:70s} VGPR {v.get('VGPRs'):4d}  spilled SGPR {v.get('SGPRs Spill', 0):4d}  scratch {v.get('ScratchSize', 0):3d}  waves/SIMD {v.get('Occupancy', 0)}  LDS {v.get('LDS Size', 0)}")
