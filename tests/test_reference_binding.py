"""The drop-in classes constructed from the REFERENCE's own objects, as main.py:34-43 constructs them:
`DatasetCls(args)` -> `ModelCls(args, dataset)`.  Build container only -- skipped where /root/reference is absent (nothing of
the reference travels); the reference is imported from where it lies, with make_golden.py's two stand-ins for packages off
the arithmetic path (tests/golden/make_golden.py header).  CPU: construction never touches the HIP engine (it is built on
first use), so an attribute rename on either side -- the members base_model.py:54-62 / ltr_models.py:44-79 read from
`params` and `dataset`, the state_dict keys, the matrix and mask layouts -- is caught here."""
import os
import shutil
import tempfile

import numpy as np
import pytest

REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'TextGCN')), reason='the reference tree is not present on this box')


@pytest.fixture(scope='module')
def ref():
    """(make_golden module -- its import installs the stand-ins and imports TextGCN --, scratch dir)"""
    import make_golden as mg
    cwd = os.getcwd()
    work = tempfile.mkdtemp(prefix='tgcn_bind_')
    yield mg, work
    os.chdir(cwd)
    shutil.rmtree(work, ignore_errors=True)


def _check_graph(model, ds, golden_idx=None, golden_val=None):
    """model.graph (CSR) == the reference dataset's coalesced norm_matrix; mask CSR == train_user_dict"""
    idx = ds.norm_matrix._indices().numpy()
    val = ds.norm_matrix._values().numpy()
    got_idx, got_val = model.graph.to_coo()
    assert np.array_equal(got_idx, idx) and np.array_equal(got_val.view(np.uint32), val.view(np.uint32))
    if golden_idx is not None:
        assert np.array_equal(got_idx, golden_idx) and np.array_equal(got_val.view(np.uint32), golden_val.view(np.uint32))
    rp, items = model._mask_rowptr_host, model._mask_items_host
    for u in range(ds.n_users):
        assert sorted(ds.train_user_dict[u]) == items[rp[u]:rp[u + 1]].tolist()
    assert model.n_users == ds.n_users and model.n_items == ds.n_items
    assert np.array_equal(model.test_users, np.sort(ds.test_df.user_id.unique()))
    assert model.user_mapping_dict == dict(ds.user_mapping[['remap_id', 'org_id']].values)
    assert model.item_mapping_dict == dict(ds.item_mapping[['remap_id', 'org_id']].values)


def test_lightgcn_from_reference_dataset_and_parser(ref, golden):
    """main.py's lgcn flow with only the registry's model class changed: parse_args -> BaseDataset(args) -> LightGCN(args, ds)"""
    mg, work = ref
    from textgcn_amd.model import LightGCN
    g1 = golden('g1_dummy')
    data = os.path.join(work, 'dummy')
    shutil.copytree(os.path.join(REF, 'data/dummy'), data)
    args = mg.run_args(['--model', 'lgcn', '--no_train', '--predict', '-k', '1', '2', '3'], data, work)
    ds = mg.TextGCN.BaseDataset(args)
    theirs = mg.TextGCN.BaseModel(args, ds)
    ours = LightGCN(args, ds)
    assert sorted(ours.state_dict().keys()) == sorted(theirs.state_dict().keys()) == ['embedding_item.weight', 'embedding_user.weight']
    for k, v in theirs.state_dict().items():
        assert ours.state_dict()[k].shape == v.shape and ours.state_dict()[k].dtype == v.dtype
    ours.load_state_dict(theirs.state_dict())            # a reference checkpoint loads as is (base_model.py:278-287)
    for name in ('k', 'lr', 'uid', 'save', 'quiet', 'epochs', 'dropout', 'emb_size', 'n_layers', 'save_path', 'batch_size', 'reg_lambda',
                 'evaluate_every', 'neg_samples', 'slurm'):      # base_model.py:33-52
        assert getattr(ours, name) == getattr(theirs, name), name
    assert ours.training is False and str(ours.device) == str(theirs.device) == 'cpu'
    assert ours.true_test_lil == theirs.true_test_lil and ours.metrics == theirs.metrics
    _check_graph(ours, ds, g1['norm_idx'], g1['norm_val'])
    # every public member main.py / subclasses reach for exists with the reference's call shape
    import inspect
    for name in ('layer_aggregation', 'layer_combination', 'score_pairwise', 'score_batchwise', 'predict', 'evaluate', 'fit', 'get_loss',
                 'bpr_loss', 'reg_loss', 'load_model', 'checkpoint', '_copy_params', '_copy_dataset_params', '_init_embeddings', '_add_vars'):
        a = list(inspect.signature(getattr(ours, name)).parameters)
        b = list(inspect.signature(getattr(theirs, name)).parameters)
        assert a[:len(b)] == b or name == 'layer_combination', (name, a, b)
    assert isinstance(type(ours).representation, property) and isinstance(type(theirs).representation, property)
    with pytest.raises(RuntimeError, match='ROCm GPU only'):
        ours.representation            # the product path fails loudly without the HIP device: no CPU fallback


def test_ltr_linear_from_reference_dataset_and_parser(ref, golden):
    """ltr_linear: parse_args -> LTRDataset(args) -> LTRLinear(args, ds) on the synthetic 60x40 LTR set of G4"""
    mg, work = ref
    import torch
    from textgcn_amd.ltr import LTRLinear, LTRLinearWPop
    g4 = golden('g4_ltr')
    data = mg.g2_synth(work, write=False)
    mg.ltr_files(data)
    args = mg.run_args(['--model', 'ltr_linear', '--no_train', '-k', '5', '10', '--batch_size', '32', '--freeze'], data, work)
    ds = mg.TextGCN.LTRDataset(args)
    theirs = mg.TextGCN.LTRLinear(args, ds)
    ours = LTRLinear(args, ds)
    assert sorted(ours.state_dict().keys()) == sorted(theirs.state_dict().keys()) == list(g4['state_keys'])
    ours.load_state_dict(theirs.state_dict())
    assert not ours.embedding_user.weight.requires_grad and not ours.embedding_item.weight.requires_grad      # --freeze
    assert ours.feature_names == theirs.feature_names
    for name in ('items_as_avg_reviews', 'users_as_avg_reviews', 'users_as_avg_desc', 'items_as_desc'):       # ltr_models.py:49-55
        assert torch.equal(getattr(ours, name).cpu(), getattr(theirs, name).cpu()), name
        assert np.array_equal(getattr(ours, name).cpu().numpy(), g4[name])
    assert list(ours.all_items) == list(theirs.all_items)
    # the instance-level rebinding of ltr_models.py:175-179
    for name in ('evaluate', 'score_pairwise', 'score_batchwise'):
        assert name in ours.__dict__ and name in theirs.__dict__
    w, b = ours.effective_weights()
    assert np.array_equal(w, theirs.layers[0].weight.detach().numpy().reshape(-1)) and b == float(theirs.layers[0].bias.detach())
    _check_graph(ours, ds)
    # ltr_pop on the same dataset object (ltr_models.py:213-241)
    args_p = mg.run_args(['--model', 'ltr_pop', '--no_train', '-k', '5', '10', '--batch_size', '32', '--freeze'], data, work)
    theirs_p = mg.TextGCN.LTRLinearWPop(args_p, ds)
    ours_p = LTRLinearWPop(args_p, ds)
    assert sorted(ours_p.state_dict().keys()) == sorted(theirs_p.state_dict().keys())
    assert ours_p.feature_names == theirs_p.feature_names and ours_p.layers[0].weight.shape == theirs_p.layers[0].weight.shape


def test_adv_sampling_from_reference_dataset_and_parser(ref):
    mg, work = ref
    from textgcn_amd.adv_sampling import AdvSamplModel
    data = mg.g2_synth(work, write=False)
    args = mg.run_args(['--model', 'adv_sampling', '--no_train', '-k', '3', '5'], data, work)
    ds = mg.TextGCN.AdvSamplDataset(args)
    theirs = mg.TextGCN.AdvSamplModel(args, ds)
    ours = AdvSamplModel(args, ds)
    assert ours.pos_samples == theirs.pos_samples == 5
    assert sorted(ours.state_dict().keys()) == sorted(theirs.state_dict().keys())
    _check_graph(ours, ds)


def test_registry_has_the_reference_names():
    """main.get_class (main.py:16-22): the same four names, [DatasetCls, ModelCls] pairs"""
    import re
    from textgcn_amd.model import get_class
    src = open(os.path.join(REF, 'main.py')).read()
    names = re.findall(r"'(\w+)': \[", src)
    assert sorted(names) == ['adv_sampling', 'lgcn', 'ltr_linear', 'ltr_pop']
    for n in names:
        pair = get_class(n)
        assert len(pair) == 2 and all(isinstance(c, type) for c in pair)
    with pytest.raises(KeyError):
        get_class('nope')
