"""Adapted-layer list of the SD1.5 UNet at 512x512 (latent 64x64), LyCORIS preset "full" (BASELINE.json configs[1]:
"LoCon dim=16 conv_dim=8 on SD1.5 UNet, bf16, batch=4 512^2").

Enumerated from the SD1.x architecture: block channels 320/640/1280/1280, two resnets per down block (three
per up block), one transformer block (depth 1) behind every resnet of the three attention levels and in the middle,
context 77 x 768, time embedding 1280.  ``proj_in`` / ``proj_out`` of Transformer2DModel are 1x1 convolutions in SD1.x
(they are nn.Linear in SDXL).  Preset "full" adapts every nn.Linear / nn.Conv2d inside Transformer2DModel, ResnetBlock2D,
Downsample2D and Upsample2D (conv_in / conv_out / time_embedding are outside those classes).

Same entry format as benchmarks/sdxl_shapes.py.
"""


def sd15_unet_layers(batch: int = 4):
    L = []

    def lin(count, M, I, O, tag, sib=1):
        L.append(dict(kind="linear", M=M, I=I, O=O, count=count, tag=tag, sib=sib))  # sib: see benchmarks/sdxl_shapes.py

    def conv(count, hw_in, C, O, k, stride, tag):
        L.append(dict(kind="conv", B=batch, C=C, H=hw_in, W=hw_in, O=O, k=k, stride=stride, pad=k // 2, count=count, tag=tag))

    def transformer(count, d, hw):
        tok = batch * hw * hw
        conv(2 * count, hw, d, d, 1, 1, f"proj_in/out 1x1 @{d}")
        lin(3 * count, tok, d, d, f"attn1 to_q/to_k/to_v @{d}", sib=3)
        lin(3 * count, tok, d, d, f"attn1 to_out + attn2 to_q/to_out @{d}")
        lin(2 * count, batch * 77, 768, d, f"attn2 to_k/to_v @{d}", sib=2)
        lin(count, tok, d, 8 * d, f"ff.net.0.proj (GEGLU) @{d}")
        lin(count, tok, 4 * d, d, f"ff.net.2 @{d}")

    def resnet(count, hw, cin, cout):
        conv(count, hw, cin, cout, 3, 1, f"resnet conv1 {cin}->{cout}@{hw}")
        conv(count, hw, cout, cout, 3, 1, f"resnet conv2 {cout}@{hw}")
        lin(count, batch, 1280, cout, f"time_emb_proj->{cout}")
        if cin != cout:
            conv(count, hw, cin, cout, 1, 1, f"shortcut {cin}->{cout}@{hw}")

    # down path
    resnet(2, 64, 320, 320); transformer(2, 320, 64); conv(1, 64, 320, 320, 3, 2, "downsample 320")
    resnet(1, 32, 320, 640); resnet(1, 32, 640, 640); transformer(2, 640, 32); conv(1, 32, 640, 640, 3, 2, "downsample 640")
    resnet(1, 16, 640, 1280); resnet(1, 16, 1280, 1280); transformer(2, 1280, 16); conv(1, 16, 1280, 1280, 3, 2, "downsample 1280")
    resnet(2, 8, 1280, 1280)
    # middle
    resnet(2, 8, 1280, 1280); transformer(1, 1280, 8)
    # up path (skip connections concatenate channels)
    resnet(3, 8, 2560, 1280); conv(1, 16, 1280, 1280, 3, 1, "upsample conv 1280@16")
    resnet(2, 16, 2560, 1280); resnet(1, 16, 1920, 1280); transformer(3, 1280, 16); conv(1, 32, 1280, 1280, 3, 1, "upsample conv 1280@32")
    resnet(1, 32, 1920, 640); resnet(1, 32, 1280, 640); resnet(1, 32, 960, 640); transformer(3, 640, 32)
    conv(1, 64, 640, 640, 3, 1, "upsample conv 640@64")
    resnet(1, 64, 960, 320); resnet(2, 64, 640, 320); transformer(3, 320, 64)
    return L


def distinct_linear_shapes(batch: int = 4):
    seen, out = set(), []
    for l in sd15_unet_layers(batch):
        if l["kind"] == "linear":
            key = (l["M"], l["I"], l["O"])
            if key not in seen:
                seen.add(key)
                out.append(key)
    return out
