"""oracle/ -- TEST INFRASTRUCTURE ONLY.

A CPU (PyTorch fp32) restatement of the hot path of vFones/situation-recognition
(`model.py`, `utils/imsitu_encoder.py`, `utils/imsitu_scorer.py`), used as the
checker for the HIP implementation in `situation_recognition_amd/`.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
may import this package.  The product package never imports it and has no CPU
fallback: it raises if the HIP extension is missing.

Pinning status (see DESIGN.md "Oracle"):
  * GGSNN / FCGGNN / losses / encoder / scorer: PINNED -- checked against golden
    vectors produced by importing the reference itself (`oracle/gen_golden.py`,
    fixtures in `tests/golden/`).
  * ResNet backbone arithmetic: PARITY UNPINNED by the reference -- it lives in
    torchvision (absent from /root/reference and from this image, version
    unpinned by the reference).  `oracle/ref_resnet.py` restates the published
    torchvision ResNet v1.5 graph; it is cross-checked structurally against
    `transformers.ResNetModel` (tests/test_oracle_resnet.py).
"""
