"""CPU oracle for the EVOKE hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

This package is a plain-PyTorch (fp32, eager, CPU) restatement of the reference
algorithm for the path BASELINE.json's north_star names.  It exists only so that
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
can check / time the reference arithmetic where ``/root/reference`` is absent
(the GPU box).  Nothing under ``evoke_amd/`` imports it; the product path fails
loudly when the HIP library is missing instead of falling back to this code.

Form: pure functions over a parameter dict whose keys are the *reference's*
``state_dict`` names, so the same weights can be loaded
into the imported reference, this oracle and the HIP engine by key.

Pinning:
  * the reference has no tests; the pins are (i) the known-answer vectors that
    were captured from the imported reference (tests/golden/kat.json), (ii)
    golden outputs regenerated from the *imported reference itself* in the build
    container by tests/golden/make_golden.py (procedural weights, seeded
    inputs), (iii) the shipped tokenizer JSON and prediction CSV header rows.
  * un-vendored third-party arithmetic: torchvision==0.16.2 ``resnet101`` is not
    installed -> the trunk is restated from the public architecture and its
    parity is pinned by structure only (42,500,160 params, state_dict keys and
    shapes); HF ``BertModel`` (reference pins transformers==4.23.1, container
    has 5.15.0) is pinned against the in-container HF implementation.
"""
