"""CPU oracle for the adversarial_learner hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / the timed CPU baseline.

PARITY UNPINNED: the reference (antonilo/unsupervised_detection) ships no
tests, golden vectors or fixtures, and its arithmetic lives in TensorFlow
1.13.1, which cannot be installed here.  The oracle therefore restates the
reference files together with the TF-1.13 kernel semantics they rely on
 and is pinned by (i) known-answer properties
derived from the reference code (tests/test_oracle_*.py) and (ii) a second,
independent explicit-loop numpy restatement of every index-math op
(oracle/oracle_np.py).
"""
