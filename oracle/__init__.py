"""CPU oracle for the MV-Former SCL training step.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch (CPU, fp32/fp64)
restatement of the reference's algorithm for the hot path named in
BASELINE.json.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it, and only as the checker /
reported baseline -- never as the thing that is measured or shipped.  The
product package (`video_rep_learning_amd`) never imports it.

Parity pinning (see DESIGN.md "Oracle"):
  * head / loss / glue (`oracle.head`, `oracle.scl`, `oracle.model`): PINNED --
    checked against golden vectors produced by importing the reference's own
    modules (`CARL_MVF/models/utils.py`, `models/mvformer.py`,
    `models/resnet_c2d.py::MLPHead`, `algos/scl.py`) in the build container
    (`tests/golden/gen_golden.py`, fixtures in `tests/golden/*.npz`).
  * ViT backbone (`oracle.vit`): the arithmetic lives in the un-vendored
    third-party dependency timm (pinned timm==0.9.2, reference README.md:24)
    which is absent offline.  It restates timm 0.9.2 `VisionTransformer`
    semantics and is cross-checked against an independent implementation
    (HuggingFace `transformers.ViTModel`, random seeded weights).  The
    reference holds no test or golden vector at this boundary, so for the ViT
    arithmetic alone: "parity unpinned by the reference" (anchored on the
    call sites CARL_MVF/models/transformer.py:59,188,322-331).
"""
