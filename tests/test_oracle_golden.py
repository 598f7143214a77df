"""Pin the CPU oracle (oracle/selfc_oracle.py) against vectors produced by the
real reference modules (tools/make_golden.py).  Bit-exact for the index
shuffles (Haar, FrequencyAnalyzer, Quantization), <=2e-6 relative for float
paths (different but equivalent summation orders of the same fp32 convs)."""
import torch

from conftest import load_golden, rel_err, seeded_fill, subdict
from oracle import selfc_oracle as O

T = 7
FTOL = 2e-6


def test_haar_bit_exact():
    g = load_golden("g1_haar")
    assert torch.equal(O.haar_fwd(g["x"]), g["y"])
    assert torch.equal(O.haar_fwd(g["y"]), g["y2"])           # second level, C=12
    assert torch.equal(O.haar_inv(g["y"]), g["xr"])
    assert torch.equal(O.haar_inv(g["zrand"]), g["zrand_inv"])  # odd sizes (5x7)
    assert abs(O.haar_jacobian(g["x"].shape, False) - float(g["jac_fwd"])) < 1e-9
    assert abs(O.haar_jacobian(g["y"].shape, True) - float(g["jac_rev"])) < 1e-9
    # rev(fwd(x)) == x to fp32 rounding
    assert (O.haar_inv(O.haar_fwd(g["x"])) - g["x"]).abs().max() < 5e-7


def test_freq_bit_exact_and_not_inverse():
    g = load_golden("g2_freq")
    assert torch.equal(O.freq_fwd(g["x"]), g["y"])
    assert torch.equal(O.freq_inv(g["z"]), g["z_rev"])
    assert torch.equal(O.freq_inv(g["y"]), g["y_rev"])
    # trap 3: rev is not the inverse of fwd
    assert (O.freq_inv(O.freq_fwd(g["x"])) - g["x"]).abs().max() > 0.1


def test_quant_bit_exact():
    g = load_golden("g9_quant")
    assert torch.equal(O.quantize(g["x"]), g["y"])


def test_denseblock():
    g = load_golden("g3_denseblock")
    for tag in ("f", "g"):
        y = O.dense_block(subdict(g, tag), g[f"{tag}_x"])
        assert rel_err(y, g[f"{tag}_y"]) < FTOL


def test_d2dt_clip_boundaries():
    g = load_golden("g4_d2dt")
    for tag in ("f", "g"):
        y = O.d2dt(subdict(g, tag), g[f"{tag}_x"], T)
        assert rel_err(y, g[f"{tag}_y"]) < FTOL
    # zero padding at clip ends: frames of clip 0 do not see clip 1
    x = g["f_x"].clone()
    y0 = O.d2dt(subdict(g, "f"), x, T)
    x[T:] += 1.0
    y1 = O.d2dt(subdict(g, "f"), x, T)
    assert torch.equal(y0[:T], y1[:T])


def _check_invblock(name, kind, t):
    g = load_golden(name)
    y, s = O.invblock(kind, g, g["x"], 3, t, rev=False)
    assert rel_err(y, g["y_fwd"]) < FTOL and rel_err(s, g["s_fwd"]) < 5e-6
    assert abs(O.invblock_jacobian(s, g["x"].shape[0], False).item() - g["jac_fwd"].item()) < 1e-3 * abs(g["jac_fwd"].item()) + 1e-4
    y, s = O.invblock(kind, g, g["x"], 3, t, rev=True)
    assert rel_err(y, g["y_rev"]) < FTOL and rel_err(s, g["s_rev"]) < 5e-6
    assert abs(O.invblock_jacobian(s, g["x"].shape[0], True).item() - g["jac_rev"].item()) < 1e-3 * abs(g["jac_rev"].item()) + 1e-4
    # invertibility
    z, _ = O.invblock(kind, g, g["x"], 3, t, rev=False)
    xr, _ = O.invblock(kind, g, z, 3, t, rev=True)
    assert rel_err(xr, g["x"]) < 1e-5


def test_invblock_dbnet():
    _check_invblock("g5_invblock_dbnet", "DBNet", T)


def test_invblock_d2dt():
    _check_invblock("g5_invblock_d2dt", "D2DTNet", T)


def test_large_stack():
    g = load_golden("g8_large_stack")
    z = O.large_fwd(g, g["x"], T)
    assert rel_err(z, g["z"]) < 1e-5
    assert float(g["loss_c"]) == 0.0
    xr = O.large_inv_from_latent(g, g["z"], T)
    assert rel_err(xr, g["x_rev"]) < 1e-5


def test_haar_net():
    g = load_golden("g8_haar_net")
    z = O.haar_net_fwd(g, g["x"], [1], T, "DBNet")
    assert rel_err(z, g["z"]) < FTOL
    assert rel_err(z[:, :3], g["lr"]) < FTOL
    assert abs((z[:, 3:] ** 2).mean().item() - g["hf_meansq"].item()) < 1e-6
    xr = O.haar_net_inv(g, g["z"], [1], T, "DBNet")
    assert rel_err(xr, g["x_rev"]) < FTOL
    assert rel_err(xr, g["x"]) < 1e-5


def test_globalagg():
    g = load_golden("g6_globalagg")
    for tag in ("a", "b"):
        y = O.global_agg(g, g[f"{tag}_x"], T)
        assert rel_err(y, g[f"{tag}_y"]) < FTOL


def test_stp_gmm_head_and_sample():
    g = load_golden("g7_stp_gmm")
    raw = O.stp_v2_parameters(g, g["lr"], T)
    assert rel_err(raw, g["raw"]) < 2e-5
    v = O.stp_v2_gmm_sample(g["raw"], g["eps"])
    assert rel_err(v, g["v"]) < 2e-6


def test_stp_l2_full_reverse():
    g = load_golden("g7_stp_l2_full_rev")
    ops = load_golden("g8_large_stack")
    stp = subdict(g, "stp_net")
    hf = O.stp_v2_parameters(stp, g["lr"], T)
    assert rel_err(hf, g["hf"]) < 2e-5
    xr = O.large_inv_from_latent(ops, torch.cat((g["lr"], hf), 1), T)
    assert rel_err(xr, g["x_rev"]) < 5e-5


def test_selfc_haar_variant_with_stp_v1():
    g = load_golden("g8_selfc_haar")
    z, loss = O.selfc_haar_fwd(g, g["x"], [1], T, "DBNet")
    assert rel_err(z, g["z"]) < FTOL
    assert abs(loss.item() - g["loss_c"].item()) < 1e-5 * abs(g["loss_c"].item()) + 1e-7
    xr, hf = O.selfc_haar_rev(g, g["lr"], [1], T, "DBNet")
    assert rel_err(hf, g["hf"]) < 1e-5
    assert rel_err(xr, g["x_rev"]) < 1e-5


def _fcb_stp_params(seed, c=32):
    """state_dict-shaped parameters of STP v1 with the FeatureCalapseBlock conditioner, filled like the fixture."""
    shapes = {}
    for blk, (ci, co) in {"blk1": (48, 192), "blk2": (192, 16 * c)}.items():
        for i in range(1, 5):
            kt = 3 if i == 1 else 1
            shapes[f"{blk}.conv{i}.weight"] = (128, ci + 128 * (i - 1), kt, 3, 3)
            shapes[f"{blk}.conv{i}.bias"] = (128,)
        shapes[f"{blk}.conv5.weight"] = (co, ci + 512, 3, 3, 3)
        shapes[f"{blk}.conv5.bias"] = (co,)
    shapes["tail.1.weight"] = (9, c, 1, 1, 1)
    shapes["tail.1.bias"] = (9,)
    params = {k: torch.empty(v) for k, v in shapes.items()}
    seeded_fill(params, seed)
    return params


def test_selfc_haar_variant_with_feature_calapse_block():
    g = load_golden("g8_selfc_haar_fcb")
    stp = _fcb_stp_params(int(g["stp_fill_seed"]))
    y1 = O.feature_calapse_block(subdict(stp, "blk1"), g["lr"], T)
    assert rel_err(y1, g["blk1_y"]) < 1e-5
    params = dict(g)
    params.update({"stp_net." + k: v for k, v in stp.items()})
    z, loss = O.selfc_haar_fwd(params, g["x"], [1], T, "DBNet")
    assert rel_err(z, g["z"]) < FTOL
    assert abs(loss.item() - g["loss_c"].item()) < 1e-4 * abs(g["loss_c"].item()) + 1e-7
    xr, hf = O.selfc_haar_rev(params, g["lr"], [1], T, "DBNet")
    assert rel_err(hf, g["hf"]) < 2e-5
    assert rel_err(xr, g["x_rev"]) < 2e-5


def test_gaussian_downsample():
    g = load_golden("g10_gauss")
    y = O.gaussian_downsample(g["x"])
    assert y.shape == g["y"].shape and rel_err(y, g["y"]) < 2e-6


def test_train_step_losses_and_gradnorm():
    """G11: the hand restatement of optimize_parameters against the step run on the reference modules."""
    g = load_golden("g11_train_step")
    params = {**{k: v for k, v in load_golden("g8_large_stack").items() if k.startswith("operations.")},
              **{k: v for k, v in load_golden("g7_stp_l2_full_rev").items() if k.startswith("stp_net.")}}
    x = load_golden("g8_large_stack")["x"]
    assert rel_err(O.gaussian_downsample(x), g["ref_l"]) < 2e-6
    p = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "haar" not in k) for k, v in params.items()}
    l_fit, l_rec, loss = O.train_step_losses(p, x, g["ref_l"], 7)
    assert abs(l_fit.item() - float(g["l_forw_fit"])) < 1e-5 * float(g["l_forw_fit"])
    assert abs(l_rec.item() - float(g["l_back_rec"])) < 1e-4 * float(g["l_back_rec"])
    loss.backward()
    names = g["names"]
    norms = torch.tensor([float(p[n].grad.norm()) for n in names], dtype=torch.float64)
    assert rel_err(norms, g["grad_norms"]) < 2e-3
    total = float(torch.sqrt((norms ** 2).sum()))
    assert abs(total - float(g["grad_norm"])) < 2e-3 * float(g["grad_norm"])


def test_multistep_lr_restart_trace():
    g = load_golden("g11_lr_trace")
    lr = O.multistep_lr_restart(1e-4, 12, [3, 6, 9], restarts=[5], weights=[0.5], gamma=0.5)
    assert rel_err(torch.tensor(lr, dtype=torch.float64), g["lr"]) < 1e-12


def test_ssim_y():
    g = load_golden("g12_ssim_y")
    got = O.ssim_per_frame(O.rgb_to_y(g["a"]), O.rgb_to_y(g["b"]))
    assert rel_err(torch.tensor(got, dtype=torch.float64), g["ssim"]) < 1e-6
    assert abs(O.ssim_per_frame(O.rgb_to_y(g["a"]), O.rgb_to_y(g["a"]))[0] - 1.0) < 1e-6


def test_irn_reverse_with_pinned_hf():
    """InvRescaleNet.forward(rev=True) with the reference's torch.rand HF tensor pinned (fixture g14)."""
    g = load_golden("g14_irn_rev")
    out = O.irn_rev(g, g["lr"], g["hf45"], [1], T)
    assert rel_err(out, g["x_rev"]) < 1e-5


def test_codec_variant_pieces_and_tiling():
    """f4: codec-variant shapes (SelfC_Codec_arch_inv.py) - InvBlockExp(15|3), narrow STP, GlobalAgg(24) over 3-frame clips,
    and forward_test's segmenting / tiling, against vectors from the reference's own modules (fixture g15)."""
    g = load_golden("g15_codec")
    y, s = O.invblock("D2DTNet", subdict(g, "operations.1"), g["blk_x"], 3, 3)
    assert rel_err(y, g["blk_y"]) < FTOL and rel_err(s, g["blk_s"]) < 5e-6
    xr, _ = O.invblock("D2DTNet", subdict(g, "operations.1"), g["blk_x"], 3, 3, rev=True)
    assert rel_err(xr, g["blk_xrev"]) < FTOL
    assert rel_err(O.global_agg(subdict(g, "stp_net.global_m1"), g["ga_x"], 3), g["ga_y"]) < FTOL
    assert rel_err(O.codec_stp_parameters(subdict(g, "stp_net"), g["stp_lr"], 3), g["stp_raw"]) < 2e-5
    enc = O.codec_encode_tiled(g, g["x"], 5)
    assert rel_err(enc, g["enc_lr"]) < 1e-5
    assert torch.equal(O.quantize(g["enc_lr"]), g["lr_q"])
    assert rel_err(O.codec_decode_tiled(g, g["lr_q"], 5), g["dec_hr"]) < 5e-5
    # seg_add_pad repeats the SECOND-TO-LAST frame (utils/util.py:340), and remove undoes add
    v = torch.arange(5.).reshape(1, 5, 1, 1, 1)
    padded, pad = O.seg_add_pad(v, 3)
    assert pad == 1 and padded.reshape(-1).tolist() == [0, 1, 2, 3, 4, 3]
    assert torch.equal(O.seg_remove_pad(padded, pad, 3), v)


def test_stp_v1_gmm_head():
    """STP v1 GMM branch (SelfC_arch_inv.py:118-128,151-163; `.cuda()` at :161 neutralised when the fixture was made)."""
    g = load_golden("g16_stp_v1_gmm")
    raw = O.stp_v1_parameters(g, g["lr"], T)
    assert rel_err(raw, g["raw"]) < 2e-5
    assert rel_err(O.stp_v1_gmm_sample(g["raw"], g["eps"]), g["v"]) < 2e-6


def test_stp_v2_gmm_thin_head():
    """fh_loss 'gmm_thin' (ReLU between the head's layers, SelfC_GMM_arch_inv.py:345-354); the chain's weights are G7's."""
    g7, g = load_golden("g7_stp_gmm"), load_golden("g17_stp_gmm_thin")
    params = {k: v for k, v in g7.items() if k.split(".")[0] in ("local_m1", "local_m2", "global_m1", "global_m2", "other_stp_modules")}
    params.update({k: v for k, v in g.items() if k.startswith("tail_gmm.")})
    raw = O.stp_v2_parameters(params, g["lr"], T, thin=True)
    assert rel_err(raw, g["raw"]) < 2e-5
    assert rel_err(O.stp_v2_gmm_sample(g["raw"], g["eps"]), g["v"]) < 2e-6
