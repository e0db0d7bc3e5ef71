"""Process-wide kernel-selection switches, the counterparts of the two cuDNN flags the reference scripts
set (/root/reference/2.distributed.py:85 ``cudnn.deterministic = True`` under ``--seed``, :122
``cudnn.benchmark = True``).

``benchmark``      the first time a convolution shape is seen, time the candidate tile configurations of the
                   implicit-GEMM kernels on the real operands and keep the fastest for that shape (per op:
                   fprop / dgrad / wgrad).  Off: the built-in shape heuristics choose.
``deterministic``  run-to-run reproducible results: the split-K reduction of wgrad goes through ordered
                   partial slabs (or a single split) instead of float atomics.  Everything else in the
                   package is order-deterministic already.
"""
benchmark = False
deterministic = False
