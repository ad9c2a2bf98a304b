"""Host-side drop-in contract: model-name grammar (main.py:41-147), checkpoint layout (main.py:522-526), and the
device-side quantise+normalise input step."""
import importlib
import os
import types

import numpy as np
import pytest
import torch


def _args(model, **kw):
    d = dict(model=model, input_channels=1, input_image_size=64, intermediate_channels=32, z_dimension=64, sigma_decoder=0.1,
             require_rsample=True, num_pixelcnn_layers=4, pixelcnn_activation="ReLu", nll=1, quantization="2", decoder_out_channels=2)
    d.update(kw)
    return types.SimpleNamespace(**d)


def test_select_model_name_grammar(pkg):
    m, mp = pkg.select_model(_args("normal_vae_1_kl_0_mmd"))
    assert (m.decoder_out_channels, m.kl, m.mmd, m.sigma_decoder, m.z_dimensions) == (1, 1.0, 0.0, 0.1, 64)
    assert mp["model_name"] == "VAE" and mp["is_decoder_out_normal"] and mp["decoder_out_channels"] == 1
    m, mp = pkg.select_model(_args("categorical_vae_0.5_kl_10_mmd", sigma_decoder=0.0))
    assert (m.decoder_out_channels, m.kl, m.mmd) == (2, 0.5, 10.0) and not mp["is_decoder_out_normal"]
    with pytest.raises(AssertionError):
        pkg.select_model(_args("normal_vae_1_kl_0_mmd", sigma_decoder=0.0))      # main.py:91-93
    with pytest.raises(AssertionError):
        pkg.select_model(_args("normal_vae_x_kl_0_mmd"))
    with pytest.raises(AssertionError):
        pkg.select_model(_args("vae"))
    m, mp = pkg.select_model(_args("pixelcnn_4"))                                # main.py:57-66
    assert mp["model_name"] == "PixelCNN" and m.only_pixelcnn and m.pixelcnn is not None and m.num_pixelcnn_layers == 4
    assert mp["decoder_out_channels"] == 0 and mp["pixelcnn_out_channels"] == 2 and not hasattr(m, "encoder")
    m, mp = pkg.select_model(_args("categorical_pixelvae_1_kl_0_mmd", sigma_decoder=0.0))
    assert mp["model_name"] == "PixelVAE" and m.pixelcnn is not None and m.decoder_out_channels == 2 and m.pixelcnn_out_channels == 2
    assert [k for k in m.state_dict()][:3] == ["pixelcnn.layers.0.weight", "pixelcnn.layers.0.bias", "pixelcnn.layers.0.mask"]
    m, mp = pkg.select_model(_args("normal_pixelvae_1_kl_0_mmd", sigma_decoder=0.0))
    assert m.decoder_out_channels == 1 and tuple(m.state_dict()["pixelcnn.layers.0.weight"].shape) == (32, 2, 7, 7)
    with pytest.raises(AssertionError):
        pkg.select_model(_args("normal_pixelvae_1_kl_0_mmd", sigma_decoder=0.1))  # main.py:91-93
    with pytest.raises(AssertionError):
        pkg.select_model(_args("pixelcnn_x"))


def test_checkpoint_layout_roundtrips_with_the_oracle_shell(pkg, oracle, tmp_path):
    """state_dict saved by this package loads into the oracle shell (same keys/shapes as the reference) and back."""
    M = importlib.import_module("moving-mnist-vae_amd.model")
    m = M.VAE(1, 32, 1, 2, 32, False, False)
    opt = torch.optim.Adam(list(m.parameters()))
    path = pkg.save_checkpoint(m, opt, 3, str(tmp_path))
    assert os.path.basename(path) == "latest-model.model"
    ck = torch.load(path, weights_only=False)
    assert set(ck) == {"epoch", "state_dict", "optimizer"} and ck["epoch"] == 3
    o = oracle.OracleVAE(1, 32, 1, 2, 32, False, False)
    o.load_state_dict(ck["state_dict"])                      # strict: identical keys and shapes
    for k, v in o.state_dict().items():
        assert torch.equal(v, m.state_dict()[k]), k
    m2 = M.VAE(1, 32, 1, 2, 32, False, False)
    assert pkg.load_checkpoint(path, m2) == 3
    for k, v in m2.state_dict().items():
        assert torch.equal(v, m.state_dict()[k]), k


def test_checkpoint_of_a_deeper_variant_names_its_shape(pkg, tmp_path):
    """blocks_per_stage travels with the checkpoint: the package loader rebuilds the deeper net from it and refuses a mismatch."""
    M = importlib.import_module("moving-mnist-vae_amd.model")
    for blocks in (2, 4):
        m = M.VAE(1, 32, 1, 2, 32, False, False, blocks_per_stage=blocks)
        opt = torch.optim.Adam(list(m.parameters()))
        path = pkg.save_checkpoint(m, opt, 1, str(tmp_path / f"b{blocks}"))
        v = pkg.checkpoint_variant(path)
        assert v["blocks_per_stage"] == blocks
        m2 = M.VAE(1, 32, 1, 2, 32, False, False, blocks_per_stage=v["blocks_per_stage"], compute_dtype=v["compute_dtype"])
        assert pkg.load_checkpoint(path, m2) == 1
        assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
        with pytest.raises(ValueError):
            pkg.load_checkpoint(path, M.VAE(1, 32, 1, 2, 32, False, False))
    a = _args("normal_vae_1_kl_0_mmd")
    a.blocks_per_stage, a.compute_dtype = 2, "f32"
    model, mp = pkg.select_model(a)
    assert model.blocks_per_stage == 2 and model.compute_dtype == "f32" and mp["blocks_per_stage"] == 2


@pytest.mark.gpu
def test_quantise_frames_matches_kmeans_predict(pkg):
    g = torch.Generator().manual_seed(0)
    frames = torch.randint(0, 256, (7, 64, 64), generator=g, dtype=torch.uint8)
    for centres in ([0.0038, 0.8808], [0.0, 0.3, 0.62, 0.97]):
        x = frames.numpy().astype(np.float32) / 255.0
        ref = np.argmin((x[..., None] - np.asarray(centres, dtype=np.float32)) ** 2, axis=-1)     # kmeans.predict
        labels, image = pkg.quantise_frames(frames.cuda(), centres, 0.0521, 0.2222)
        assert np.array_equal(labels.cpu().numpy(), ref)
        np.testing.assert_allclose(image.cpu().numpy(), (ref.astype(np.float32) - 0.0521) / 0.2222, rtol=1e-6)


def _kmeans_fixture(q):
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"kmeans_q{q}.npz"))
    return d["frames"], d["centres"], d["labels"].astype(np.int64), float(d["data_mean"]), float(d["data_std"])


@pytest.mark.parametrize("q", [2, 4])
def test_kmeans_fixture_is_nearest_centre(q):
    """CPU: the committed scikit-learn outputs (oracle/make_kmeans_fixture.py: KMeans.fit / .predict as utils.py:287 and main.py:25
    call them) are what the device kernel's rule -- nearest centre on the ToTensor scale, lowest index on ties -- gives."""
    frames, centres, labels, mean, std = _kmeans_fixture(q)
    x = frames.astype(np.float32) / 255.0
    ref = np.argmin((x[..., None] - centres.astype(np.float32)) ** 2, axis=-1)
    assert np.array_equal(ref, labels)
    assert abs(labels.mean() - mean) < 1e-4 and abs(labels.std() - std) < 1e-4          # utils.py:296-305 (rounded to 4 digits)


@pytest.mark.gpu
@pytest.mark.parametrize("q", [2, 4])
def test_quantise_frames_matches_sklearn_kmeans_predict(pkg, q):
    """The device-side input step against scikit-learn's own KMeans.predict on the same frames (fixture: data only)."""
    frames, centres, labels, mean, std = _kmeans_fixture(q)
    got, image = pkg.quantise_frames(torch.from_numpy(frames).cuda(), centres, mean, std)
    assert np.array_equal(got.cpu().numpy(), labels)
    np.testing.assert_allclose(image.cpu().numpy(), (labels.astype(np.float32) - np.float32(mean)) / np.float32(std), rtol=1e-6)


def test_clips_from_npz_array_layout(pkg):
    """The reference reads (N, C, W, H), transposes to (N, H, W, C) (movingmnistdataset.py:15) and ToTensor makes each sample
    (C, H, W): restated with numpy here, step by step, and compared with the one-transpose form the loader uses."""
    rng = np.random.default_rng(0)
    arr = rng.integers(0, 256, size=(3, 20, 8, 6), dtype=np.uint8)              # (N, C, W, H), non-square on purpose
    dataset_view = arr.transpose(0, 3, 2, 1)                                    # dataset: (N, H, W, C)
    to_tensor = np.stack([s.transpose(2, 0, 1) for s in dataset_view])          # ToTensor on an HWC ndarray: (C, H, W)
    got = pkg.clips_from_npz_array(arr)
    assert got.dtype == torch.uint8 and tuple(got.shape) == (3, 20, 6, 8)
    assert np.array_equal(got.numpy(), to_tensor)
    with pytest.raises(ValueError):
        pkg.clips_from_npz_array(arr.astype(np.float32))


@pytest.mark.gpu
def test_prepare_batch_stages_the_input_once(pkg, oracle):
    """prepare_batch (main.py:383-387) writes the f32 image AND its storage-type copy into the workspace in one pass; the forward that
    follows skips its conversion pass only for exactly that tensor.  uint8 labels (an eighth of the int64 transport) give the same image;
    a modified or foreign input goes through the conversion again; results are identical to the bit."""
    M = importlib.import_module("moving-mnist-vae_amd.model")
    dev = torch.device("cuda")
    labels = oracle.synthetic_labels(6, 64, seed=21).view(6, 4096)
    torch.manual_seed(2)
    m = M.VAE(1, 32, 1, 2, 32, False, False, compute_dtype="bf16").to(dev).train()
    m.injected_eps = torch.randn(6, 32, 1, 1, device=dev)
    image, target = m.prepare_batch(labels, dev, oracle.DATA_MEAN, oracle.DATA_STD, False)
    assert target is image and m._staged is not None
    ref = ((labels.float() - oracle.DATA_MEAN) / oracle.DATA_STD).view(6, 1, 64, 64)
    assert (image.cpu() - ref).abs().max().item() <= 1e-6
    out_staged = [t.detach().clone() for t in m(image)]
    assert m._staged is None                                  # consumed
    out_plain = [t.detach().clone() for t in m(image.clone())]       # a different tensor: converted by the forward itself
    for a, b in zip(out_staged, out_plain):
        assert torch.equal(a, b)
    image8, _ = m.prepare_batch(labels.to(torch.uint8), dev, oracle.DATA_MEAN, oracle.DATA_STD, False)
    assert torch.equal(image8, image)
    image8.mul_(2.0)                                          # touched after staging: the staged copy is stale and must not be used
    out_mod = m(image8)
    out_ref = m((image * 2.0))
    for a, b in zip(out_mod, out_ref):
        assert torch.equal(a, b)
    imgc, tgt = m.prepare_batch(labels.to(torch.uint8), dev, oracle.DATA_MEAN, oracle.DATA_STD, True)
    assert tgt.dtype == torch.int64 and tuple(tgt.shape) == (6, 64, 64)


@pytest.mark.gpu
def test_prepare_batch_identity_and_alignment(pkg, oracle):
    """The staged-input shortcut is keyed on the tensor OBJECT (a weak reference), not on (pointer, version): a fresh tensor that happens to
    get the freed image's block back from the caching allocator is converted like any other input.  Label pointers need no alignment:
    labels[1:] of an int64 batch (8-byte aligned) and an odd uint8 offset normalise to the same values as the aligned copy."""
    M = importlib.import_module("moving-mnist-vae_amd.model")
    L = importlib.import_module("moving-mnist-vae_amd._lib")
    dev = torch.device("cuda")
    labels = oracle.synthetic_labels(4, 64, seed=9).view(4, 4096)
    torch.manual_seed(2)
    m = M.VAE(1, 32, 1, 2, 32, False, False, compute_dtype="bf16").to(dev).train()
    m.injected_eps = torch.randn(4, 32, 1, 1, device=dev)
    image, _ = m.prepare_batch(labels, dev, oracle.DATA_MEAN, oracle.DATA_STD, False)
    ptr0 = image.data_ptr()
    ref = [t.detach().clone() for t in m(image.clone())]
    image, _ = m.prepare_batch(labels, dev, oracle.DATA_MEAN, oracle.DATA_STD, False)
    del image                                            # the staged tensor dies; its block goes back to the allocator
    other = torch.empty((4, 1, 64, 64), device=dev)      # ... and most likely comes back here
    other.copy_(ref[0].new_zeros(()).expand_as(other) + 0.25)
    out = m(other)
    exp = m(torch.full((4, 1, 64, 64), 0.25, device=dev))
    for a, b in zip(out, exp):
        assert torch.equal(a, b), ("the forward ran on a stale staged copy", other.data_ptr() == ptr0)
    # misaligned label views
    lib = L.lib()
    flat = labels.to(dev).view(-1)
    st = torch.cuda.current_stream().cuda_stream
    for lab in (flat[1:4097 + 4096], flat.to(torch.uint8)[3:4096 + 3]):
        n = lab.numel()
        assert lab.data_ptr() % 16 != 0
        img = torch.empty(n, device=dev)
        if lab.dtype == torch.int64:
            L.check(lib.mmvae_normalise_labels(L.ptr(lab), n, float(oracle.DATA_MEAN), float(oracle.DATA_STD), L.ptr(img), st), "normalise")
            assert torch.equal(img, (lab.float() - oracle.DATA_MEAN) / oracle.DATA_STD) or \
                (img - (lab.float() - oracle.DATA_MEAN) / oracle.DATA_STD).abs().max().item() <= 1e-6


def test_binding_checks_the_abi_version(pkg):
    """include/mmvae.h asks bindings to compare mmvae_abi_version() with MMVAE_ABI_VERSION at load time: _lib.lib() does."""
    import re
    L = importlib.import_module("moving-mnist-vae_amd._lib")
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "mmvae.h")).read()
    assert int(re.search(r"#define MMVAE_ABI_VERSION (\d+)", hdr).group(1)) == L.ABI_VERSION
    lib = L.lib()
    assert lib.mmvae_abi_version() == L.ABI_VERSION
    saved, L._lib, L.ABI_VERSION = L._lib, None, L.ABI_VERSION + 1
    try:
        with pytest.raises(L.MmvaeError):
            L.lib()
    finally:
        L._lib, L.ABI_VERSION = saved, L.ABI_VERSION - 1


@pytest.mark.gpu
def test_moving_mnist_clips_loader(pkg, tmp_path):
    """npz on disk -> device-resident clips -> batches of k-means labels shaped like the reference loader's (B, C*H*W)."""
    rng = np.random.default_rng(1)
    arr = rng.integers(0, 256, size=(10, 20, 64, 64), dtype=np.uint8)
    np.savez(tmp_path / "movingmnisttrain.npz", arr)
    centres = [0.0038, 0.8808]
    loader = pkg.MovingMNISTClips(str(tmp_path), centres, batch_size=4, device="cuda", shuffle=False)
    assert len(loader) == 3 and len(loader.train_data) == 10
    batches = list(loader)
    assert [tuple(b.shape) for b in batches] == [(4, 81920), (4, 81920), (2, 81920)] and batches[0].dtype == torch.int64
    x = arr.transpose(0, 1, 3, 2).astype(np.float32) / 255.0                     # ToTensor scale, (N, C, H, W)
    ref = np.argmin((x[..., None] - np.asarray(centres, dtype=np.float32)) ** 2, axis=-1).reshape(10, -1)
    assert np.array_equal(torch.cat(batches).cpu().numpy(), ref)
    shuffled = pkg.MovingMNISTClips(arr, centres, batch_size=10, device="cuda", shuffle=True, seed=3)
    (b,) = list(shuffled)
    assert sorted(map(tuple, b.cpu().numpy()[:, :64].tolist())) == sorted(map(tuple, ref[:, :64].tolist()))


@pytest.mark.gpu
@pytest.mark.parametrize("capturable", [False, True])
def test_fused_adam_resume(pkg, oracle, tmp_path, capturable):
    """Save after 2 steps, resume into a fresh model/optimiser, third step equals an uninterrupted run -- also with the step count on
    the device (capturable=True: state_dict() reads the device counter, load_flat_state() restores it)."""
    M = importlib.import_module("moving-mnist-vae_amd.model")
    dev = torch.device("cuda")
    args = types.SimpleNamespace(data_ratio_of_labels=None, dataset="MovingMNIST", quiet=True)
    batches = [oracle.synthetic_labels(8, 64, seed=40 + i).view(8, 4096) for i in range(3)]
    noise = [(torch.randn(8, 32, 1, 1, generator=torch.Generator().manual_seed(i)), torch.randn(8, 32, generator=torch.Generator().manual_seed(9 + i))) for i in range(3)]

    def run(model, opt, idx):
        for i in idx:
            model.injected_eps, model.injected_true_samples = noise[i][0].to(dev), noise[i][1].to(dev)
            pkg.train(model, [batches[i]], opt, dev, args, data_mean=oracle.DATA_MEAN, data_std=oracle.DATA_STD)

    torch.manual_seed(1)
    a = M.VAE(1, 32, 1, 2, 32, False, False, compute_dtype="f32").to(dev)
    oa = M.FusedAdam(list(a.parameters()), capturable=capturable)
    init = {k: v.clone() for k, v in a.state_dict().items()}
    run(a, oa, [0, 1, 2])
    b = M.VAE(1, 32, 1, 2, 32, False, False, compute_dtype="f32")
    b.load_state_dict(init); b.to(dev)
    ob = M.FusedAdam(list(b.parameters()), capturable=capturable)
    run(b, ob, [0, 1])
    path = pkg.save_checkpoint(b, ob, 0, str(tmp_path))
    saved = torch.load(path, weights_only=False)["optimizer"]["state"]
    assert all(float(e["step"]) == 2.0 for e in saved.values()), "the checkpoint must carry the real step count"
    c = M.VAE(1, 32, 1, 2, 32, False, False, compute_dtype="f32").to(dev)
    oc = M.FusedAdam(list(c.parameters()), capturable=capturable)
    pkg.load_checkpoint(path, c, oc, map_location=dev)
    assert oc._t == ob._t == 2 and torch.equal(oc._m, ob._m) and torch.equal(oc._v, ob._v) and torch.equal(c._flat, b._flat)
    if capturable:
        assert int(oc._step_dev.item()) == 2
    run(c, oc, [2])
    torch.cuda.synchronize()
    # the step is bit-reproducible (tests/test_boundary_gpu.py::test_train_step_is_bit_reproducible), so a resumed run IS the
    # uninterrupted run
    for k, p in c.named_parameters():
        assert torch.equal(p.detach(), dict(a.named_parameters())[k].detach()), k


def test_repr_is_the_reference_text(pkg):
    """model.py:408-439, pixelcnn is None branch."""
    M = importlib.import_module("moving-mnist-vae_amd.model")
    m = M.VAE(1, 32, 1, 2, 32, False, False, 4, "ReLu", 1, 1, 0, True, 0.1, 64)
    assert repr(m) == ("We are using an encoder which takes input of 64x64x1 and encodes into 32 dimensional latent space."
                       " Where Z is rsampled from a Normal Distribution. \nIt is then pushed into a decoder which outputs an image of"
                       " dimension 64x64x1.\nWe assume p(x/z) follows a normal distribution with mean x_recon and sigma 0.1.\n")
    m2 = M.VAE(1, 32, 2, 2, 128, False, False, 4, "ReLu", 1, 1, 0, False, 0.0, 32)
    assert repr(m2) == ("We are using an encoder which takes input of 32x32x1 and encodes into 128 dimensional latent space."
                        " \nIt is then pushed into a decoder which outputs an image of dimension 32x32x2.\n"
                        "We assume p(x/z) follows a categorical distribution. \n")


def test_package_exports(pkg):
    for name in ("VAE", "FusedAdam", "GradSync", "Communicator", "train", "select_model", "save_checkpoint", "load_checkpoint"):
        assert getattr(pkg, name) is not None
    assert not hasattr(pkg, "DataParallelTrainer")


def test_bench_gpus_flag_is_not_silently_ignored():
    """`bench.py --gpus N` started plainly must launch N ranks or fail: with fewer GPUs than N it refuses before touching a GPU."""
    import subprocess
    import sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    n = torch.cuda.device_count() + 1
    r = subprocess.run([_sys.executable, os.path.join(root, "bench.py"), "--gpus", str(max(n, 2))], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)
