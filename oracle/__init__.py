"""CPU oracle for the dynesty bounding + proposal hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
product: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it, and only as the checker.  The product
(``dynesty_amd``) never imports this package and fails loudly when the HIP
library is missing.

The oracle is a NumPy/SciPy restatement of the reference algorithms
(``/root/reference/py/dynesty/bounding.py`` and ``internal_samplers.py``;
every function cites the file:line it follows).  It deliberately calls the same
third-party numerics the reference calls (``scipy.linalg.eigh``,
``scipy.cluster.vq.kmeans2``, ``numpy.random.Generator(PCG64)``), in the same
order, so that it is *bit-identical* to the reference on the same inputs.

Parity pinning: ``tests/golden/*.npz`` were produced by importing the real
reference in the build container (``tools/make_golden.py``) and
``tests/test_oracle_golden.py`` checks the oracle against them bit-for-bit;
``tests/test_oracle_vs_reference.py`` additionally runs the oracle side by side
with the live reference whenever ``/root/reference`` is present.
"""
