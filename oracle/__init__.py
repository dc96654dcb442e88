"""CPU oracle for the FBS hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  The shipped solver (``fasta_python_amd``) never imports it and has no CPU fallback:
it raises when the HIP library is missing.

Contents
  fasta_np.py   NumPy restatement of the reference solver loop, linear maps, prox operators and
                stop rules (reference: fasta/__init__.py, linalg.py, proximal.py, stopping.py).
  problems.py   Restated example recipes (sparse least squares, NNLS, l1-ball LASSO, TV dual)
                plus the splitmix/Irwin-Hall synthetic generator twin used by bench.py.
  make_golden.py  Runs the *real* reference core (importable from /root/reference in the build
                container only) and writes tests/golden/*.npz.  Never runs on the GPU box.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks fasta_np bit-for-bit against
fixtures captured from the reference's own ``fasta.fasta`` (see make_golden.py).  The reference
ships no tests or golden vectors of its own, so those captured fixtures are
the pin.
"""
