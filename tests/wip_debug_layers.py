"""Debug helper (not collected by pytest): per-layer comparison of the HIP engine's saved
activations / gradients with a float64 torch evaluation on the GPU."""
import sys
import torch

sys.path.insert(0, ".")
from cinema_amd import _lib, engine as E  # noqa
from cinema_amd.models.siren import Siren  # noqa


def run(precision, N=300, H=128, lat=32, L=3, mod=(1, 3), out=11):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = Siren(3, lat, out, H, L, 30, 30, True, list(mod), precision=precision).to(dev)
    st = net.hot_state(dev)
    eng = st.engine
    d = eng.desc
    coords = (torch.rand(N, 3, device=dev) * 2 - 1)
    latent = torch.randn(N, lat, device=dev) * 0.05
    coords.requires_grad_(True)
    latent.requires_grad_(True)
    y = net((coords, latent))
    gout = torch.randn_like(y)
    y.backward(gout)
    # float64 reference
    P = {k: v.detach().double() for k, v in net.state_dict().items()}
    x = coords.detach().double()
    lt = latent.detach().double()
    xs, cs = [], []
    x.requires_grad_(True)
    lt.requires_grad_(True)
    h = x
    acts = []
    for i in range(L + 1):
        z = h @ P[f"net.{i}.linear.weight"].t() + P[f"net.{i}.linear.bias"]
        if i in mod:
            stt = lt @ P[f"net.{i}.linear_lats.weight"].t() + P[f"net.{i}.linear_lats.bias"]
            a = 30 * z * stt[:, :H] + stt[:, H:]
        else:
            a = 30 * z
        h = torch.sin(a)
        h.retain_grad()
        acts.append((h, torch.cos(a)))
    o = torch.cat((h, lt), -1) @ P["final_linear.weight"].t() + P["final_linear.bias"]
    o.backward(gout.double())
    npad = E.round_up(N, 128)
    sv = eng.ws.saved

    def rel(a, b):
        return float((a.double() - b).abs().max() / b.abs().max())

    print(f"--- precision {precision}")
    for i in range(L + 1):
        yi = sv[d.sv_y[i], :N, :H]
        ci = sv[d.sv_c[i], :N, :H]
        print(f"layer {i}: y err {rel(yi, acts[i][0].detach()):.2e}  c err {rel(ci, acts[i][1].detach()):.2e}")
    print(f"out err {rel(y.detach(), o.detach()):.2e}")
    if precision == 0:
        i = 2
        yi = sv[d.sv_y[i], :N, :H].double(); ci = sv[d.sv_c[i], :N, :H].double()
        ph = torch.atan2(yi, ci)
        ref_a = torch.atan2(acts[i][0].detach(), acts[i][1].detach())
        bad = ((ph - ref_a).abs() > 1e-3)
        print("layer2 bad fraction", float(bad.double().mean()))
        print("bad per column block of 16:", [float(bad[:, k:k+16].double().mean()) for k in range(0, H, 16)])
        print("bad per row block of 16:", [round(float(bad[k:k+16].double().mean()),2) for k in range(0, N, 16)])
        print("bad rows in first 32:", bad[:32].any(dim=1).int().tolist())
        print("bad cols in first 32:", bad[:, :32].any(dim=0).int().tolist())
        h1 = acts[1][0].detach()
        zref = h1 @ P["net.2.linear.weight"].t() + P["net.2.linear.bias"]
        print("ref 30z[0,:4]", (30*zref[0,:4]).tolist())
        print("hip  phase mod[0,:4]", ph[0,:4].tolist(), " ref phase mod", ref_a[0,:4].tolist())
        print("z mod 2pi [0,:4]", torch.atan2(torch.sin(zref[0,:4]), torch.cos(zref[0,:4])).tolist())
    gb = eng.ws.gbuf
    for i in range(L, -1, -1):
        # dz_i = d loss / d z_i ; reference: acts[i][0].grad * cos * omega (* s)
        pass
    print(f"dcoords err {rel(coords.grad, x.grad):.2e}  dlat err {rel(latent.grad, lt.grad):.2e}")
    for k, p in net.named_parameters():
        ref = None
    # parameter grads via a second float64 pass with params requiring grad
    P2 = {k: v.detach().double().requires_grad_(True) for k, v in net.state_dict().items()}
    h = coords.detach().double()
    for i in range(L + 1):
        z = h @ P2[f"net.{i}.linear.weight"].t() + P2[f"net.{i}.linear.bias"]
        if i in mod:
            stt = latent.detach().double() @ P2[f"net.{i}.linear_lats.weight"].t() + P2[f"net.{i}.linear_lats.bias"]
            a = 30 * z * stt[:, :H] + stt[:, H:]
        else:
            a = 30 * z
        h = torch.sin(a)
    o2 = torch.cat((h, latent.detach().double()), -1) @ P2["final_linear.weight"].t() + P2["final_linear.bias"]
    o2.backward(gout.double())
    for k, p in net.named_parameters():
        print(f"  grad {k}: {rel(p.grad, P2[k].grad):.2e}")


if __name__ == "__main__":
    for prec in (_lib.PREC_F32, _lib.PREC_BF16):
        run(prec)
