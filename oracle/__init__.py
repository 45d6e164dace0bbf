"""CPU oracle for the Footprints hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the shipped product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and there only as the checker / the CPU reference timing -- never as the
thing measured or shipped.  The product path (``footprints_amd``) never
imports this package and fails loudly when its HIP library is missing.

Parity status (see DESIGN.md section "Oracle"):
  * decoder, heads, output concat, loss: PINNED against the reference's own
    code (``/root/reference/footprints/network.py``, ``training/losses.py``)
    through the committed fixtures in ``tests/golden/`` (generated here by
    ``tests/golden/make_golden.py`` which imports the reference).
  * encoder: the reference delegates to ``torchvision.models.resnet34``
    (torchvision 0.4.2, environment.yml:9) which is NOT present under
    /root/reference nor installed in this image => that part is
    "parity unpinned" by the reference; it is restated from the published
    ResNet-34 architecture and pinned against stock ``torch.nn`` ops.
"""
