#!/usr/bin/env python3
"""Registers, LDS, scratch, occupancy and code size of the kernels in a hipcc --save-temps .s file:  tools/kernel_resources.py FILE.s [REGEX]"""
import re
import sys

txt = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)^; Occupancy: (\d+)", txt, re.S | re.M):
    name, body, occ = m.group(1), m.group(2), m.group(3)
    if pat and not pat.search(name):
        continue
    g = lambda k: (re.search(r"; %s[:=] *(\d+)" % k, body) or [None, "?"])[1]
    print("%-44s vgpr %3s sgpr %3s lds %6s scratch %3s occupancy %2s code %6s B" % (re.sub(r"^_ZN4sphx\d+", "", name)[:44], g("NumVgprs"), g("TotalNumSgprs"), g("LDSByteSize"), g("ScratchSize"), occ, g("codeLenInByte ")))
