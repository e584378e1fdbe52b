"""Adapted-layer list of the SDXL UNet at 1024x1024 (latent 128x128), batch 1, LyCORIS preset "full".

diffusers / checkpoints are not available offline, so the workload is the *shape list* enumerated from the SDXL
architecture: block channels 320/640/1280, transformer depth 0/2/10, context dim 2048 x 77 tokens,
time embedding 1280.  788 adapted layers = 739 nn.Linear + 49 nn.Conv2d.

Each entry: dict(kind="linear", M, I, O, count) or dict(kind="conv", B, C, H, W, O, k, stride, pad, count)
(H, W = INPUT spatial size).
"""


def sdxl_unet_layers(batch: int = 1):
    L = []

    def lin(count, M, I, O, tag, sib=1):
        # sib = n: every n consecutive instances read the SAME tensor (to_q / to_k / to_v of a self-attention take the block's hidden
        # state, to_k / to_v of a cross-attention the text context): what the host model's call pattern gives the adapters
        L.append(dict(kind="linear", M=M * batch, I=I, O=O, count=count, tag=tag, sib=sib))

    def conv(count, hw_in, C, O, k, stride, tag):
        L.append(dict(kind="conv", B=batch, C=C, H=hw_in, W=hw_in, O=O, k=k, stride=stride, pad=k // 2, count=count, tag=tag))

    # transformer blocks at 32x32 (1024 tokens, d=1280): 60 blocks (10 per Transformer2DModel x 6)
    lin(180, 1024, 1280, 1280, "attn1 to_q/to_k/to_v @1280", sib=3)
    lin(192, 1024, 1280, 1280, "attn1 to_out, attn2 to_q/to_out, proj_in/out @1280")
    lin(60, 1024, 1280, 10240, "ff.net.0.proj (GEGLU) @1280")
    lin(60, 1024, 5120, 1280, "ff.net.2 @1280")
    # transformer blocks at 64x64 (4096 tokens, d=640): 10 blocks (2 per Transformer2DModel x 5)
    lin(30, 4096, 640, 640, "attn1 to_q/to_k/to_v @640", sib=3)
    lin(40, 4096, 640, 640, "attn1 to_out, attn2 to_q/to_out, proj_in/out @640")
    lin(10, 4096, 640, 5120, "ff.net.0.proj @640")
    lin(10, 4096, 2560, 640, "ff.net.2 @640")
    # cross-attention K/V from the 77 x 2048 text context
    lin(120, 77, 2048, 1280, "attn2 to_k/to_v @1280", sib=2)
    lin(20, 77, 2048, 640, "attn2 to_k/to_v @640", sib=2)
    # ResnetBlock2D.time_emb_proj (one row)
    lin(5, 1, 1280, 320, "time_emb_proj->320")
    lin(5, 1, 1280, 640, "time_emb_proj->640")
    lin(7, 1, 1280, 1280, "time_emb_proj->1280")
    # 3x3 stride-1 convs of the resnets / upsamplers
    conv(7, 128, 320, 320, 3, 1, "resnet conv 320@128")
    conv(6, 64, 640, 640, 3, 1, "resnet conv 640@64")
    conv(10, 32, 1280, 1280, 3, 1, "resnet conv 1280@32")
    conv(1, 64, 320, 640, 3, 1, "resnet conv1 320->640")
    conv(1, 32, 640, 1280, 3, 1, "resnet conv1 640->1280")
    conv(1, 32, 1920, 1280, 3, 1, "up resnet conv1 1920->1280")
    conv(1, 64, 1920, 640, 3, 1, "up resnet conv1 1920->640")
    conv(1, 64, 1280, 640, 3, 1, "up resnet conv1 1280->640")
    conv(1, 64, 960, 640, 3, 1, "up resnet conv1 960->640")
    conv(1, 128, 960, 320, 3, 1, "up resnet conv1 960->320")
    conv(1, 64, 1280, 1280, 3, 1, "upsample conv 1280@64")
    conv(1, 128, 640, 640, 3, 1, "upsample conv 640@128")
    conv(2, 32, 2560, 1280, 3, 1, "up resnet conv1 2560->1280")
    conv(2, 128, 640, 320, 3, 1, "up resnet conv1 640->320")
    # 3x3 stride-2 downsamplers
    conv(1, 128, 320, 320, 3, 2, "downsample 320")
    conv(1, 64, 640, 640, 3, 2, "downsample 640")
    # 1x1 shortcuts
    conv(1, 64, 320, 640, 1, 1, "shortcut 320->640")
    conv(1, 32, 640, 1280, 1, 1, "shortcut 640->1280")
    conv(2, 32, 2560, 1280, 1, 1, "shortcut 2560->1280")
    conv(1, 32, 1920, 1280, 1, 1, "shortcut 1920->1280")
    conv(1, 64, 1920, 640, 1, 1, "shortcut 1920->640")
    conv(1, 64, 1280, 640, 1, 1, "shortcut 1280->640")
    conv(1, 64, 960, 640, 1, 1, "shortcut 960->640")
    conv(1, 128, 960, 320, 1, 1, "shortcut 960->320")
    conv(2, 128, 640, 320, 1, 1, "shortcut 640->320")
    assert sum(l["count"] for l in L if l["kind"] == "linear") == 739
    assert sum(l["count"] for l in L if l["kind"] == "conv") == 49
    return L


def layer_rows(l):
    """(M, I_eff, O): rows of the activation matrix the adapter sees, features in / out (im2col for k > 1)."""
    if l["kind"] == "linear":
        return l["M"], l["I"], l["O"]
    ho = (l["H"] + 2 * l["pad"] - l["k"]) // l["stride"] + 1
    return l["B"] * ho * ho, l["C"] * l["k"] * l["k"], l["O"]


def algorithmic_bytes(l, algo="lokr", esize=2, factor=8, rank=16):
    """activation traffic of one adapted layer, fwd + bwd, plus 3 passes over the fp32 factors.
    fwd: read x, write delta;  bwd: read g, read x, write dx.  Conv: x is counted once per pass (C*H*W), not im2col'd."""
    if l["kind"] == "linear":
        M, I, O = l["M"], l["I"], l["O"]
        x_elems, y_elems = M * I, M * O
        kk = 1
    else:
        ho = (l["H"] + 2 * l["pad"] - l["k"]) // l["stride"] + 1
        x_elems, y_elems = l["B"] * l["C"] * l["H"] * l["W"], l["B"] * l["O"] * ho * ho
        I, O, kk = l["C"], l["O"], l["k"] * l["k"]
    act = esize * (3 * x_elems + 2 * y_elems)
    if algo == "lokr":
        nparam = factor * factor + (I // factor) * (O // factor) * kk
    elif algo == "locon":
        nparam = rank * (I * kk + O)
    elif algo == "loha":
        nparam = 2 * rank * (I * kk + O)
    else:
        nparam = O
    return act + 3 * 4 * nparam
