"""CPU oracle: a plain-PyTorch (fp32, CPU) functional restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and only as
the checker (or as the timed CPU baseline), never as the thing shipped.  The product package
has no CPU fallback and raises when the HIP library is missing.

Parity pin: every function here is checked against fixtures under ``tests/golden/`` that were
produced by importing the reference itself (``/root/reference``, via
``tests/golden/make_golden.py``) in the build container -- see ``tests/test_oracle_golden.py``.
The arithmetic below the reference (conv2d, instance_norm, interpolate, ...) lives in PyTorch
ATen (reference pin: pytorch 1.2/1.4, requirements.txt:84 / README.md:24; here torch 2.10 CPU);
the ops used have unchanged semantics between those versions.

Each function cites the reference file:line it follows.
"""
