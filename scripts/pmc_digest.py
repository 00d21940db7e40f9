"""Digest of a scripts/pmc_wgrad.sh counter dump: per kernel instance the ratios the round-3 verdict asked about — LDS bank-conflict
cycles / LDS-active cycles, VALU instructions per MFMA, MFMA-busy share of the kernel, VALU / MFMA co-execution.
usage: pmc_digest.py <dump.txt>"""
import re, sys
txt = open(sys.argv[1]).read()
for blk in txt.split("('")[1:]:
    name = blk.split("'")[0]
    grid = blk.split("'")[2] if blk.count("'") > 2 else ""
    d = {m.group(1): float(m.group(2)) for m in re.finditer(r"(\w+)\s+(\d+)\n", blk)}
    if d.get("SQ_LDS_IDX_ACTIVE", 0) <= 0:
        continue
    busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    gui = d.get("GRBM_GUI_ACTIVE", 0.0)
    print(f"{name:40s} grid {grid:>8s}  LDS conflict/active {d['SQ_LDS_BANK_CONFLICT'] / d['SQ_LDS_IDX_ACTIVE']:.3f}  VALU per MFMA {d['SQ_INSTS_VALU'] / max(1.0, d.get('SQ_INSTS_MFMA', 0)):.2f}  "
          f"MFMA-busy / (GUI/8 x 1024 SIMDs) {busy / (gui / 8 * 1024) if gui else 0:.3f}  VALU||MFMA co-exec / MFMA-busy {d.get('SQ_VALU_MFMA_COEXEC_CYCLES', 0) / max(1.0, busy):.3f}  "
          f"LDS-active / wave-cycles {d['SQ_LDS_IDX_ACTIVE'] / max(1.0, d.get('SQ_WAVE_CYCLES', 0)):.3f}")
