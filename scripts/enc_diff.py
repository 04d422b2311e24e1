import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden")
import numpy as np, torch, recipe
from posteriflow_amd import npe
g = np.load("tests/golden/encoder.npz")
enc = npe.LeanStrainEncoder()
shapes = {k: v.shape for k, v in enc.state_dict().items() if k != "pos.pe"}
enc.load_state_dict(recipe.fill_state_dict(shapes, seed=103), strict=False)
enc = enc.eval().cuda()
strain = recipe.strain_batch(4, 3, seed=7).cuda()
torch.backends.cuda.matmul.allow_tf32 = False
with torch.no_grad():
    ctx = enc(strain).cpu().numpy()
    clean = enc._sanitize(strain)
    stem = enc._stem(clean).transpose(1, 2)[:2].cpu().numpy()
d = np.abs(ctx - g["det3_ctx"])
print("ctx max abs", d.max(axis=1), "scale", np.abs(g["det3_ctx"]).max())
print("stem max abs", np.abs(stem - g["det3_stem_out"]).max(), "scale", np.abs(g["det3_stem_out"]).max())
