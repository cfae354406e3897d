#!/usr/bin/env python3
"""Build gate: no kernel of libhalo_hip.so may use scratch (private segment) memory.

Why: launch sequences are replayed as hipGraphs (msm.hip msm_enqueue_batch).  On ROCm 7.2 a kernel with
scratch inside a replayed graph faulted once the queue's scratch had been re-assigned by other work
(round 1: k_msm_reduce1 carried 12 B/lane of spill at 256 VGPRs; DESIGN.md section 4 "Launch graphs").
k_msm_accumulate sits at the 256-VGPR cap, so one compiler or code change is enough to spill again:
the build fails instead.  Input: the -Rpass-analysis=kernel-resource-usage remarks of every translation
unit (written next to the objects by the Makefile).  Output: _obj/kernel_resources.json.
"""
import json
import re
import sys


def parse(path):
    kernels, cur = {}, None
    for line in open(path, errors="replace"):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass-analysis", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    return kernels


def main(argv):
    out, paths = argv[1], argv[2:]
    allk, bad = {}, []
    for p in paths:
        for name, r in parse(p).items():
            allk[name] = r
            if r.get("ScratchSize [bytes/lane]", "0") != "0" or r.get("Dynamic Stack", "False") != "False":
                bad.append((name, r))
    json.dump(allk, open(out, "w"), indent=1, sort_keys=True)
    if not allk:
        print("check_resources: no kernel remarks found", file=sys.stderr)
        return 1
    for name, r in bad:
        print("check_resources: kernel %s uses scratch memory (%s B/lane, dynamic stack %s, %s VGPRs): "
              "graph replay is not safe with scratch -- reduce register pressure" %
              (name, r.get("ScratchSize [bytes/lane]"), r.get("Dynamic Stack"), r.get("VGPRs")), file=sys.stderr)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
