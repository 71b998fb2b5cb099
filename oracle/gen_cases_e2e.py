"""Seeds and parse records of the whole-ref fixture (tests/golden/e2e_tiny.npz): shared by the generator
(oracle/gen_golden.py: gen_e2e_tiny) and its test (tests/test_gpu_e2e.py).  Test infrastructure."""
E2E_CASES = [  # (image seed, H, W, text seed, [(dirflag, relaflag, n_other nouns)] per sentence, ground-truth seed)
    (301, 160, 200, 8101, [("left", "left", 1), ("none", "big", 0), ("middle", "none", 2)], 9101),
    (302, 120, 176, 8102, [("right", "within", 1), ("none", "none", 0)], 9102),
    (304, 144, 152, 8103, [("none", "small", 2), ("left", "up", 1), ("middle", "right", 0)], 9103),
]
