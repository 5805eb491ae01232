import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def run(var):
    import torch
    from objectcentricocccompletion_amd.graph import GraphedStep
    from objectcentricocccompletion_amd.occ_encoder import SubMOccEncoder, synthetic_object_grids
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    model = SubMOccEncoder(grouped_points=True).to(dev)
    B = 4
    xyz, feats, bidx = synthetic_object_grids(B, 500, seed=3, device=dev)
    if 'nograd_pre' in var:
        with torch.no_grad():
            n = model(xyz, feats, bidx, B).features.shape[0]
    d = torch.zeros(xyz.shape[0], 128, dtype=torch.bfloat16, device=dev)
    if 'nonzero' in var:
        d[:1500] = (torch.randn(1500, 128, device=dev) / 1500).to(torch.bfloat16)

    def fwd_bwd():
        model.zero_grad(set_to_none=True)
        out = model(xyz, feats, bidx, B, static=True)
        out.features.backward(d)
        return out
    if 'eager_first' in var:
        o = fwd_bwd()
        if 'keep' in var:
            keep = o
        ge = [p.grad.clone() for p in model.parameters()]
    g = GraphedStep(fwd_bwd, warmup=2)
    g.replay(); torch.cuda.synchronize()
    print('OK', var, flush=True)

if __name__ == '__main__':
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for s in ['plain', 'nonzero', 'nograd_pre', 'eager_first', 'eager_first_keep', 'nograd_pre_eager_first_keep_nonzero']:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), s], capture_output=True, text=True)
            print(f'{s:40s} rc={r.returncode} {"OK" if "OK " + s in r.stdout else "FAIL"}', flush=True)
