"""Gradients of the fused Stage-1 kernel for two builds of the library, compared per parameter block."""
import sys, os, subprocess, numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, '.')
    import torch, dpd_amd
    from dpd_amd import _lib as L
    from dpd_amd.data import ofdm_rows
    from dpd_amd.models import TDNNGeneratorQAT
    dev = torch.device('cuda:0'); torch.manual_seed(0)
    gen = TDNNGeneratorQAT().to(dev)
    B, S = 64, 1029
    iq = ofdm_rows(B, S, seed=1234, device=dev); tgt = iq[:, 5:, :].contiguous()
    params = [gen.fc1.weight, gen.fc1.bias, gen.fc2.weight, gen.fc2.bias, gen.fc3.weight, gen.fc3.bias]
    wq = torch.empty(1554, device=dev); absmax = torch.zeros(8, device=dev)
    grads = torch.empty(1554, device=dev); loss = torch.zeros(1, device=dev)
    st = L.stream()
    L.call("dpd_gen_prepare_weights", *(L.fptr(p.detach()) for p in params), 0, L.fptr(wq), L.fptr(absmax), st)
    _, wsp, wsn = L.workspace(dev)
    L.call("dpd_gen_mse_step", L.fptr(iq), B, S, L.fptr(tgt), L.fptr(wq), 0.2, 0, L.fptr(absmax), L.fptr(grads), L.fptr(loss), wsp, wsn, 2, st)
    torch.cuda.synchronize()
    np.save(sys.argv[2], np.concatenate([grads.cpu().numpy(), loss.cpu().numpy()]))
    sys.exit(0)
outs = []
for lib in sys.argv[1:3]:
    f = "/tmp/g_%s.npy" % os.path.basename(lib)
    subprocess.check_call([sys.executable, __file__, "child", f], env=dict(os.environ, DPD_HIP_LIBRARY=os.path.abspath(lib)))
    outs.append(np.load(f))
a, b = outs
blocks = dict(W1=(0, 960), b1=(960, 992), W2=(992, 1504), b2=(1504, 1520), W3=(1520, 1552), b3=(1552, 1554), loss=(1554, 1555))
for k, (lo, hi) in blocks.items():
    d = np.abs(a[lo:hi] - b[lo:hi]).max(); m = np.abs(a[lo:hi]).max()
    print(f"{k:5s} max|a|={m:.3e} max|a-b|={d:.3e}", a[lo:lo+3], b[lo:lo+3])
