"""CPU oracle for the detector hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, in torch-CPU / numpy, the arithmetic of the reference's
detector math (`/root/reference/LightningFunc/{losses,accuracy}.py`, the anchor
generators under `LightningFunc/utils/` and the per-model NMS methods).  Every
function cites the reference file:line it follows.

Rules (enforced by `tests/test_no_oracle_in_product.py`):
  * only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
    `bench.py` may import anything from here;
  * the product package `objectdetectionpl_amd/` never imports it and has no CPU
    fallback -- it raises when the HIP library is missing.

Pinning: the oracle is checked bit-for-bit (index/mask/anchor work) and to
<=1e-6 rel (fp32 sums) against golden vectors that `oracle/gen_golden.py`
captured by importing the reference itself in the build container
(`tests/golden/*.npz`; the reference's Python never travels).
"""
