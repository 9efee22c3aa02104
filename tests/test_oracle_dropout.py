"""oracle/dropout.py (the numpy restatement of libigcn's dropout-mask generator) on the CPU: known answers — values that
``tests/test_gpu_ops.py::test_dropout_masks_are_a_pure_function_of_counter_and_index`` found equal to the kernel's, bit for
bit, on an MI355X (round 5) — and the statistics a mask needs."""
import numpy as np

from oracle import dropout as OD


def test_known_answers():
    u = OD.uniforms(5, 8)
    assert u.dtype == np.float32
    assert u.tolist() == [0.9984018802642822, 0.5936101078987122, 0.8251862525939941, 0.6147086024284363,
                          0.2441442608833313, 0.6102808117866516, 0.6006595492362976, 0.809348464012146]
    assert OD.uniforms((1 << 40) + 12345, 4).tolist() == [0.7643221616744995, 0.006993472576141357, 0.2771338224411011,
                                                          0.6378827691078186]
    m = OD.masks([((3, 5), 0.4), ((6,), 0.5)], 7)
    k = np.float32(1.0) / (np.float32(1.0) - np.float32(0.4))
    assert m[0].tolist() == [[k, k, 0.0, 0.0, k], [k, k, 0.0, k, k], [k, k, 0.0, 0.0, 0.0]]
    assert m[1].tolist() == [0.0, 2.0, 0.0, 0.0, 2.0, 0.0]        # (the second site starts at element 16, not 15)


def test_statistics_and_independence_of_counters():
    u = OD.uniforms(123, 1 << 18)
    assert 0.0 <= float(u.min()) and float(u.max()) < 1.0
    assert abs(float(u.mean()) - 0.5) < 3e-3 and abs(float((u < 0.3).mean()) - 0.3) < 3e-3
    v = OD.uniforms(124, 1 << 18)
    assert abs(float(np.corrcoef(u, v)[0, 1])) < 0.01              # consecutive steps draw unrelated masks
    assert abs(float(np.corrcoef(u[:-1], u[1:])[0, 1])) < 0.01     # neighbours are unrelated
    a, b = OD.masks([((64, 300), 0.4)], 9)[0], OD.masks([((64, 300), 0.4)], 9)[0]
    assert np.array_equal(a, b)                                    # a pure function of (counter, index, p)
    assert set(np.unique(a).tolist()) == {0.0, float(np.float32(1.0) / (np.float32(1.0) - np.float32(0.4)))}
