"""Known answers of the unlearnable-example ingress, derived by hand from the text of the reference
(/root/reference/dataset.py), not from any restatement of it:

    :1235  perturb_tensor = perturb_tensor.mul(255).clamp_(-255, 255)...numpy()   (in the dtype of the .pt file)
    :534   mask = np.zeros((h, w, c), np.float32)
    :550   mask[...] = noise                                                       (rounded to float32 here)
    :1241  self.data = self.data.astype(np.float32)
    :1287  self.data[idx] = self.data[idx] + noise                                 (float32 + float32)
    :1288  self.data[idx] = np.clip(self.data[idx], a_min=0, a_max=255)
    :1289  self.data = self.data.astype(np.uint8)                                  (truncation toward zero)

Every perturbation value below is a dyadic rational (or a stated hexadecimal double), so that each step can be followed
on paper: 255 * p is exact in the file's dtype unless the row says otherwise.  The reference itself cannot be executed in
the build container (dataset.py imports torchvision, which is absent), so these rows are what pins the ingress kernel and
the oracle's ``poison_u8`` to the reference.

Row = (dtype of the perturbation file, clean uint8 value, perturbation value, expected uint8, derivation)."""

ROWS = [
    ("float32", 0, 0.0, 0, "0 + 0"),
    ("float32", 100, 0.5, 227, "255*0.5 = 127.5; 100 + 127.5 = 227.5; trunc -> 227"),
    ("float32", 100, -0.5, 0, "100 - 127.5 = -27.5; clip -> 0"),
    ("float32", 255, -0.5, 127, "255 - 127.5 = 127.5; trunc -> 127"),
    ("float32", 1, 0.25, 64, "255*0.25 = 63.75; 1 + 63.75 = 64.75; trunc -> 64"),
    ("float32", 0, 2.0, 255, "255*2 = 510; clamp -> 255; 0 + 255 = 255"),
    ("float32", 255, -2.0, 0, "clamp(-510) = -255; 255 - 255 = 0"),
    ("float32", 1, 1.0, 255, "1 + 255 = 256; clip -> 255"),
    ("float32", 10, 0.00390625, 10, "255/256 = 0.99609375; 10.99609375; trunc -> 10"),
    ("float32", 10, -0.00390625, 9, "10 - 0.99609375 = 9.00390625; trunc -> 9"),
    ("float32", 254, 0.001953125, 254, "255/512 = 0.498046875; 254.498046875; trunc -> 254"),
    ("float32", 253, 0.01171875, 255, "255*3/256 = 2.98828125; 255.98828125; clip -> 255"),
    ("float32", 255, -0.99609375, 0, "255*(-255/256) = -254.00390625; 255 - 254.00390625 = 0.99609375; trunc -> 0"),
    ("float32", 0, -0.00390625, 0, "-0.99609375; clip -> 0"),
    ("float32", 200, 0.03125, 207, "255/32 = 7.96875; 207.96875; trunc -> 207"),
    ("float32", 200, -0.03125, 192, "200 - 7.96875 = 192.03125; trunc -> 192"),
    # float64 files (perturbation{it}.pt of main_vae_disentangle.py:91-92 is float64): product and clamp in double, ONE
    # rounding to float32 at :550
    ("float64", 100, 0.5, 227, "as the float32 row: every step exact"),
    ("float64", 10, -0.00390625, 9, "as the float32 row"),
    # p = (1 - 2^-26) / 255 rounded to double = 0x1.010100c0c0c0cp-8.  In double 255*p = 1 - 1.49e-8 (to within 2e-16);
    # the float32 neighbours of that are 1 - 5.96e-8 and 1, the midpoint is 1 - 2.98e-8, so :550 stores 1.0 and the
    # pixel becomes 0 + 1.0 = 1.  (Rounding p itself to float32 first gives 0x1.010100p-8 = (1 - 5.9e-8)/255 -- the
    # float32 below 1/255 -- whose float32 product with 255 is 0.99999994: truncated to 0.  That is the one-grey-level
    # difference a float32-first implementation makes.)
    ("float64", 0, float.fromhex("0x1.010100c0c0c0cp-8"), 1, "see above"),
    ("float64", 255, -float.fromhex("0x1.010100c0c0c0cp-8"), 254, "255 - 1.0 = 254"),
]


def as_arrays():
    """(clean uint8 [n_dtype,32,32,3], perturbations {dtype: tensor [1,3,32,32]}, expected {dtype: list}, positions): the
    rows of each dtype laid along the first image row, channel 0."""
    import numpy as np
    import torch
    out = {}
    for dt in ("float32", "float64"):
        rows = [r for r in ROWS if r[0] == dt]
        x = np.full((1, 32, 32, 3), 7, dtype=np.uint8)
        p = torch.zeros(1, 3, 32, 32, dtype=getattr(torch, dt))
        for i, (_, clean, pv, _, _) in enumerate(rows):
            x[0, i // 32, i % 32, 0] = clean
            p[0, 0, i // 32, i % 32] = pv
        want = x.copy()
        for i, (_, _, _, expect, _) in enumerate(rows):
            want[0, i // 32, i % 32, 0] = expect
        out[dt] = (x, p, want)
    return out
