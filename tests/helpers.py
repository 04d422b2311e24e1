"""Shared test helpers (oracle <-> product plumbing)."""
import torch


def oracle_state_for_product(ref):
    """The product module carries every flow weight under two prefixes, like the
    reference (flows.py:529, 532): add the flow._transform.* aliases."""
    sd = dict(ref.state_dict())
    sd.update({"flow._transform." + k[len("transform."):]: v
               for k, v in ref.state_dict().items() if k.startswith("transform.")})
    return sd


def make_pair(D, C, H, L, K, tb, scale=1.0, seed=0, device="cuda"):
    """(oracle fp32, oracle fp64, product on `device`) sharing the same weights."""
    from oracle.flow_ref import NSFPosteriorFlowRef, scale_final_layers
    from posteriflow_amd import NSFPosteriorFlow
    torch.manual_seed(seed)
    ref = NSFPosteriorFlowRef(D, C, H, L, K, tb, temperature_scale=1.0)
    scale_final_layers(ref, scale)
    ref64 = NSFPosteriorFlowRef(D, C, H, L, K, tb, temperature_scale=1.0).double()
    ref64.load_state_dict(ref.state_dict())
    flow = NSFPosteriorFlow(D, C, H, L, K, tb, temperature_scale=1.0, use_masked_context=False)
    flow.load_state_dict(oracle_state_for_product(ref))
    return ref, ref64, flow.to(device)


def flow_inputs(B, D, C, tb, seed=1, tails=True):
    """x ~ U(-1,1)^D with 2 % of entries from U(-1.2 tb, 1.2 tb) (tail branch) and a
    few exactly on / beyond the bound."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, D, generator=g) * 2 - 1
    if tails and B > 0:
        m = torch.rand(B, D, generator=g) < 0.02
        x = torch.where(m, (torch.rand(B, D, generator=g) * 2 - 1) * 1.2 * tb, x)
        x[0, 0] = tb
        if B > 1:
            x[1, D - 1] = -tb
        if B > 2:
            x[2, 0] = 1.5 * tb
    ctx = torch.randn(B, C, generator=g) if C > 0 else None
    return x, ctx
