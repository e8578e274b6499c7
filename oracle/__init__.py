"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the GP-MPC rollout hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / the reported CPU baseline.  The product
package (``gaussian_process_mpc_amd``) never imports it and has no CPU fallback.

Parity status: PINNED.  ``tests/golden/*.npz`` were produced by importing the
reference itself (``tests/golden/gen_golden.py``, run in the build container
where ``/root/reference`` is mounted) and ``tests/test_oracle_golden.py`` checks
every function below against them.
"""
