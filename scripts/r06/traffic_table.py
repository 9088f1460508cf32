"""profiles/r06_traffic_by_family.md from the eager PMC summaries (profiles/r06_{z,beam,65536}_pmc_summary.json): HBM bytes read
(FETCH_SIZE x 2: gfx950 counts 64 B per 128 B request, MI355X_MICROARCH.md) and written (WRITE_SIZE) per kernel family and pass."""
import json
import os

P = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'profiles')
GEMM = ('linear_kernel', 'mlp_chain', 'mlp_group', 'mlp_rows')
out = ["# HBM traffic by kernel family (round 6; `scripts/r06/traffic_table.py` over the eager `rocprofv3 --pmc` passes)", "",
       "Same table as round 5's (`profiles/r05_traffic_by_family.md`), on this round's kernels.  Nothing about the traffic was changed",
       "in round 6 (review item 7 was 'only after items 1-3'): the writers are still the layer outputs themselves (the pooled rows of",
       "the group / chain kernels, the per-point first-layer sums and aggregation outputs of `linear_kernel`), each written once and",
       "read once by the next launch; 23.8 MB per scene in the bench run (`roofline.traffic_bytes_per_scene`) = 0.36 TB/s at 15.2 k",
       "scenes/s, 4.5 % of the HBM peak.", ""]
for tag, label in (('z', 'benchmark scenes, eager 32-scene pass'), ('beam', 'ray-cast scenes, eager 32-scene pass'),
                   ('65536', '65536-point scenes, eager 8-scene pass')):
    s = json.load(open(os.path.join(P, 'r06_%s_pmc_summary.json' % tag)))
    rows = [(k, v['FETCH_SIZE'] * 1024 * 2 / 1e6, v['WRITE_SIZE'] * 1024 / 1e6, v.get('dispatches_per_step'))
            for k, v in s.items() if not k.startswith('_') and 'FETCH_SIZE' in v]
    rows.sort(key=lambda r: -(r[1] + r[2]))
    tr, tw = sum(r[1] for r in rows), sum(r[2] for r in rows)
    n = s['_derived']['linear_kernel']['scenes_per_step']
    out += ["## %s" % label, "", "| kernel family | launches | HBM read MB | HBM write MB |", "|---|---|---|---|"]
    out += ["| `%s` | %.0f | %.1f | %.1f |" % (k, d, r, w) for k, r, w, d in rows]
    out += ["| **all kernels** | | **%.1f** | **%.1f** |" % (tr, tw), "",
            "Per scene: %.1f MB read + %.1f MB written = %.1f MB over all kernels; the GEMM family alone %.1f MB." % (
                tr / n, tw / n, (tr + tw) / n, sum(r[1] + r[2] for r in rows if r[0] in GEMM) / n), ""]
open(os.path.join(P, 'r06_traffic_by_family.md'), 'w').write('\n'.join(out))
print('\n'.join(out[:16]))
