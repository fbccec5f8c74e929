"""CPU: host-side logic around the hot path (datasets, early stopping, driver CLI)."""
import gzip
import os
import struct
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT


def test_toy_dataset_filters_and_indexing():
    from vargp_amd.datasets import ToyDataset
    torch.manual_seed(0)
    ds = ToyDataset()
    assert ds.data.shape == (200, 2) and ds.targets.shape == (200,) and len(ds) == 200
    assert torch.equal(torch.unique(ds.targets), torch.arange(4))
    ds.filter_by_class([2, 3])
    assert len(ds) == 100
    x, y = ds[torch.arange(5)]                    # tensor-indexable, as VARGP.create_clf needs
    assert x.shape == (5, 2) and set(y.tolist()) <= {2, 3}
    ds.filter_by_class(None)
    assert len(ds) == 200


def test_early_stopper_matches_reference_semantics():
    from vargp_amd.train_utils import EarlyStopper, compute_bwt
    st = EarlyStopper(patience=2, delta=1e-4)
    st(0.5, 'a')
    st(0.50005, 'b')          # not better by delta -> counter 1
    assert st.info() == 'a' and not st.is_done()
    st(0.6, 'c')              # better -> reset
    assert st.info() == 'c'
    st(0.6, 'd'); st(0.59, 'e')
    assert st.is_done() and st.info() == 'c'
    never = EarlyStopper(patience=-1)
    for i in range(5):
        never(0.1, i)
    assert not never.is_done()
    acc = np.array([[0.9, 0.0], [0.8, 0.7]])
    assert abs(compute_bwt(acc) - (-0.1)) < 1e-12


def test_mnist_idx_parser_and_synthetic_fallback(tmp_path):
    from vargp_amd.datasets import SplitMNIST, PermutedMNIST, load_mnist
    imgs = (np.arange(6 * 28 * 28) % 256).astype(np.uint8).reshape(6, 28, 28)
    labels = np.array([0, 1, 2, 3, 4, 5], dtype=np.uint8)
    with gzip.open(tmp_path / 'train-images-idx3-ubyte.gz', 'wb') as f:
        f.write(struct.pack('>IIII', 0x00000803, 6, 28, 28) + imgs.tobytes())
    with open(tmp_path / 'train-labels-idx1-ubyte', 'wb') as f:
        f.write(struct.pack('>II', 0x00000801, 6) + labels.tobytes())
    x, y = load_mnist(str(tmp_path), train=True)
    assert x.shape == (6, 784) and torch.equal(y, torch.arange(6)) and abs(x[0, 5].item() - 5 / 255.) < 1e-7
    with pytest.raises(FileNotFoundError):
        load_mnist(str(tmp_path), train=False, synthetic=False)
    ds = SplitMNIST(str(tmp_path / 'nowhere'), train=True, n_synth=500)     # synthetic surrogate
    assert ds.data.shape == (500, 784) and 0.0 <= ds.data.min() and ds.data.max() <= 1.0
    ds.filter_by_idx(torch.arange(100))
    ds.filter_by_class([0, 1])
    assert len(ds) == 20
    pm = PermutedMNIST(str(tmp_path / 'nowhere'), train=True, n_synth=50)
    perm = torch.randperm(784)
    before = pm.data.clone()
    pm.set_task(perm)
    assert torch.equal(pm.data, before[:, perm])
    with pytest.raises(AssertionError):
        pm.set_task(perm)


def test_driver_cli_defaults_match_reference():
    sys.path.insert(0, os.path.join(ROOT, 'experiments'))
    import importlib
    drv = importlib.import_module('vargp')
    seen = {}
    for cmd, fn in [('toy', 'toy'), ('s-mnist', 'split_mnist'), ('p-mnist', 'permuted_mnist')]:
        orig = getattr(drv, fn)
        setattr(drv, fn, lambda a, c=cmd: seen.__setitem__(c, a))
        try:
            drv.main([cmd])
        finally:
            setattr(drv, fn, orig)
    assert (seen['toy'].epochs, seen['toy'].M, seen['toy'].lr, seen['toy'].beta) == (5000, 20, 1e-2, 1.0)
    assert (seen['s-mnist'].epochs, seen['s-mnist'].M, seen['s-mnist'].lr, seen['s-mnist'].beta) == (500, 60, 3e-3, 10.0)
    assert (seen['p-mnist'].epochs, seen['p-mnist'].M, seen['p-mnist'].lr, seen['p-mnist'].beta,
            seen['p-mnist'].n_tasks) == (1000, 100, 3.7e-3, 1.64, 10)
    assert seen['toy'].batch_size == 512 and seen['toy'].ep_var_mean is True


def test_toy_dataset_reproduces_the_reference_draw():
    """ToyDataset under a fixed seed == the reference's ToyDataset under the same seed (fixture written by
    tests/golden/make_golden.py from var_gp/datasets.py:21-51): same random stream, same class layout."""
    import numpy as np
    import torch
    from conftest import GOLDEN
    from vargp_amd.datasets import ToyDataset
    g = np.load(os.path.join(GOLDEN, 'toy_dataset.npz'))
    for seed in (1, 7):
        torch.manual_seed(seed)
        ds = ToyDataset()
        np.testing.assert_allclose(ds.data.numpy(), g[f'x_seed{seed}'], rtol=0, atol=1e-6)
        np.testing.assert_array_equal(ds.targets.numpy(), g[f'y_seed{seed}'])
    torch.manual_seed(3)
    ds = ToyDataset(N_K=20)
    np.testing.assert_allclose(ds.data.numpy(), g['x_nk20_seed3'], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(ds.targets.numpy(), g['y_nk20_seed3'])
