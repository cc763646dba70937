"""Markdown table of the per-kernel counters printed by tools/pmc_passes.sh (groups 0, 3, 8): MFMA-busy, LDS-array activity, bank-conflict
share and the split of wave life (parked / issue-stalled / issuing).  Usage: python tools/pmctable.py gpurun_out/<tag>_pmc.txt [top]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
top = int(sys.argv[2]) if len(sys.argv) > 2 else 26
ker, cur = collections.OrderedDict(), None
for l in txt.splitlines():
    m = re.match(r'== (.*)', l)
    if m:
        cur = m.group(1).replace('void ', '').replace('vdx::', '').strip()
        ker[cur] = {}
        continue
    m = re.match(r'\s+(\S+)\s+n=\s*(\d+) avg=\s*([\d.]+)', l)
    if m and cur:
        ker[cur][m.group(1)] = (int(m.group(2)), float(m.group(3)))
rows = []
for k, d in ker.items():
    if 'GRBM_GUI_ACTIVE' not in d or 'SQ_WAVE_CYCLES' not in d:
        continue
    cyc = d['GRBM_GUI_ACTIVE'][1] / 8                        # summed over the 8 XCDs
    mf = d.get('SQ_VALU_MFMA_BUSY_CYCLES', (0, 0))[1] / 1024 / cyc    # summed over 1024 SIMDs
    lds = d.get('SQ_LDS_IDX_ACTIVE', (0, 0))[1] / 256 / cyc  # summed over 256 CUs
    bc = d.get('SQ_LDS_BANK_CONFLICT', (0, 0))[1] / max(d.get('SQ_LDS_IDX_ACTIVE', (0, 1))[1], 1)
    wc = d['SQ_WAVE_CYCLES'][1]
    rows.append((cyc * d['GRBM_GUI_ACTIVE'][0], k, d['GRBM_GUI_ACTIVE'][0], cyc, mf, lds, bc, d['SQ_WAIT_ANY'][1] / wc,
                 d['SQ_WAIT_INST_ANY'][1] / wc, d['SQ_ACTIVE_INST_ANY'][1] / wc))
rows.sort(reverse=True)
print('| kernel | launches sampled | kernel cycles (GUI_ACTIVE / 8) | MFMA busy | LDS array active | bank-conflict share of LDS cycles | '
      'wave life parked (`SQ_WAIT_ANY`) | issue-stalled (`SQ_WAIT_INST_ANY`) | issuing |')
print('|---|---|---|---|---|---|---|---|---|')
for r in rows[:top]:
    print(f'| `{r[1]}` | {r[2]} | {r[3] / 1e3:.0f} k | {100 * r[4]:.0f} % | {100 * r[5]:.0f} % | {100 * r[6]:.0f} % | {100 * r[7]:.0f} % | '
          f'{100 * r[8]:.0f} % | {100 * r[9]:.0f} % |')
