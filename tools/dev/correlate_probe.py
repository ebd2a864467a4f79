import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from uforecon_amd import frustum
from uforecon_amd.scene import make_correlate_case
from oracle import frustum_oracle as FO
DEV = "cuda:0"
for name in ("stage1_small", "edge"):
    c = make_correlate_case(name)
    g = np.load(os.path.join(ROOT, "tests", "golden", f"correlate_{name}.npz"))
    sim, agg = frustum.correlate(c["ref_fea"].to(DEV), torch.stack(c["src_feas"]).to(DEV), c["ref_proj_pair"], c["src_proj_pairs"], c["depth_values"].to(DEV), c["view_weights"].to(DEV))
    ref = torch.from_numpy(g["similarity"])
    d = (sim.cpu() - ref).abs()
    print(name, "max abs diff", float(d.max()), "scale", float(ref.abs().max()), "n>1e-6", int((d > 1e-6).sum()), "of", d.numel())
    sims_host, _ = FO.correlate(c["ref_fea"], c["src_feas"], c["ref_proj_pair"], c["src_proj_pairs"], c["depth_values"], c["view_weights"])
    print("   oracle on this host vs golden: max", float((sims_host - ref).abs().max()))
    rel = frustum.relative_projections(c["ref_proj_pair"], c["src_proj_pairs"])
    ref_new = FO.fold_projection(c["ref_proj_pair"])
    rel_o = torch.stack([FO.relative_projection(FO.fold_projection(pp), ref_new)[:3, :4].reshape(12) for pp in c["src_proj_pairs"]])
    print("   rel proj equal to oracle's:", torch.equal(rel, rel_o))
    i = int(d.reshape(-1).argmax()); idx = np.unravel_index(i, d.shape)
    print("   worst at", idx, float(sim.cpu().reshape(-1)[i]), float(ref.reshape(-1)[i]))
    zero_mismatch = ((sim.cpu() == 0) != (ref == 0)).sum()
    print("   zero-pattern mismatches", int(zero_mismatch))
