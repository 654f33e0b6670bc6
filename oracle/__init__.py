"""CPU oracle for the hot path (TEST INFRASTRUCTURE ONLY).

This package restates, on the CPU, the arithmetic of the reference's hot path
(abeytheo/gan-inpainting: lib/models/networks.py UnetGenerator / PatchGANDiscriminator,
lib/models/loss.py, the per-batch step sequences of experiment_list/*.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
The product path (gan-inpainting_amd/) never imports anything from here and fails
loudly when the HIP library is missing.

Parity pin: tests/golden/*.npz were produced by tests/golden/make_golden.py, which
imports the unmodified reference lib/models/networks.py in the build container and
records its outputs; tests/test_oracle_golden.py checks this oracle against them.
"""
