"""Image losses (python/losses.py:4-42) as torch ops on (H,W,C) tensors."""
import torch


def l2(img, ref_img):
    return ((img - ref_img) ** 2).mean()


def l1(img, ref_img):
    return (img - ref_img).abs().mean()


def mape(img, ref_img):
    denom = (1e-2 + ref_img.mean(dim=-1, keepdim=True)).abs()
    return ((img - ref_img).abs() / denom).mean()


def downsample(img):
    """python/losses.py:14-31: same-size 2x2 box average with clamped +1 neighbours.  The
    reference indexes the flat buffer with x as the ROW coordinate (`idx = y*shape[0]*C + x*C`
    on an arange(shape[0]) x arange(shape[1]) meshgrid); for the square images every config
    uses this is the plain clamped 2x2 average, which is what is computed here."""
    h, w = img.shape[0], img.shape[1]
    yi = torch.clamp(torch.arange(h, device=img.device) + 1, max=h - 1)
    xi = torch.clamp(torch.arange(w, device=img.device) + 1, max=w - 1)
    return 0.25 * (img + img[yi] + img[:, xi] + img[yi][:, xi])


def multiscale(img, ref_img, loss_fn=l1, levels=4):
    loss = loss_fn(img, ref_img)
    for _ in range(levels - 1):
        img, ref_img = downsample(img), downsample(ref_img)
        loss = loss + loss_fn(img, ref_img)
    return loss / levels


def multiscale_l1(img, ref_img, levels=4):
    return multiscale(img, ref_img, l1, levels)
