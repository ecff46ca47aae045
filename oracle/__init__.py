"""CPU oracle for the adversarial_learner hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / the timed CPU baseline.

PARITY: pinned at the composition level, TF kernels restated.  The reference
(antonilo/unsupervised_detection) ships no tests, golden vectors or fixtures,
and its arithmetic lives in TensorFlow 1.13.1, which cannot be installed here.
The reference's own Python does run, though: oracle/make_golden.py imports
/root/reference's models/*.py (incl. AdversarialLearner.build_train_graph) on
the eager TF-1.13 stand-in oracle/tf1_shim.py and commits the outputs as
tests/golden/*.npz; tests/test_golden_reference.py holds this oracle (and the
HIP path) to them.  The TF C++ kernels behind the stand-in's primitives are
restated and checked against TensorFlow's own
unit-test vectors where the semantics are subtle: legacy / align-corners
bilinear and nearest resize (image_ops_test.py), SAME-padded strided conv2d
(conv_ops_test.py), SAME conv2d_transpose (conv2d_transpose_test.py) --
tests/test_oracle.py.  Inference BN is a closed formula.
Further pins: (i) known-answer properties derived from the reference code
(tests/test_oracle_*.py) and (ii) a second, independent explicit-loop numpy
restatement of every index-math op (oracle/oracle_np.py).
"""
