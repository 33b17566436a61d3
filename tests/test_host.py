"""Host-side logic that needs no GPU: the C-ABI library loads and exports every symbol include/facenet_hip.h declares,
the reference-API mirrors (Config, LearningRateScheduler, ImageProcessing), the flat parameter layout, and the
bucket construction for data parallelism."""
import os
import re

import pytest

from facenet_amd import _lib, parallel
from facenet_amd.config import Config, LoadConfigError, load_config
from facenet_amd.engine import Network
from facenet_amd.facenet import ImageProcessing, LearningRateScheduler, inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "facenet_hip.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(fn_\w+)\s*\(", header, flags=re.M))
    assert len(declared) >= 28
    lib = _lib.load()
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in include/facenet_hip.h but not exported"
    assert set(_lib.EXPORTS) == declared
    assert lib.fn_abi_version() == 1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.FacenetHipError, match="no CPU fallback"):
        _lib.load()


def test_network_refuses_cpu_device():
    with pytest.raises(_lib.FacenetHipError):
        Network(embedding_size=128, device="cpu")


def test_config_mirror():
    c = Config({"a": {"b": 1}, "x": 2})
    assert c.a.b == 1 and c.x == 2
    assert not c.missing and not c.missing.deeper           # config.py:85-89: missing attribute -> empty falsy Config
    assert c.as_dict == {"a": {"b": 1}, "x": 2} and c.exists("x") and not c.exists("y")
    assert repr(c) == "a: \n   b: 1\nx: 2\n"                # config.py:71-83: nested sections three spaces deeper
    assert dict(c.items())["x"] == 2 and isinstance(dict(c.items())["a"], Config)
    c.a.z = {"k": 3}                                         # assignment wraps dicts like the constructor (config.py:64-68)
    assert c.a.z.k == 3 and c.as_dict["a"] == {"b": 1, "z": {"k": 3}}
    import copy
    d = copy.deepcopy(c)
    d.a.b = 9
    assert c.a.b == 1 and d.a.b == 9 and not Config() and not Config({})
    cfg = load_config()
    assert cfg.batch_size == 100 and cfg.image.size == 160 and cfg.image.normalization == 0   # apps/configs/config.yaml:7,11,15
    assert cfg.train.epoch.size == 1000 and cfg.train.epoch.nrof_epochs == 300                # train_softmax.yaml:37 ; config.py:181-182
    with pytest.raises(LoadConfigError):
        load_config("/nonexistent/file.yaml")


def test_learning_rate_scheduler_and_image_processing():
    sched = LearningRateScheduler(Config({"value": None, "schedule": [[100, 0.05], [200, 0.005], [300, 0.0005]]}))
    assert [sched(e) for e in (0, 99, 100, 250, 300, 999)] == [0.05, 0.05, 0.005, 0.0005, 0.0005, 0.0005]
    assert LearningRateScheduler(Config({"value": 0.01, "schedule": [[1, 9.0]]}))(0) == 0.01
    img = Config({"size": 160, "normalization": 0})
    assert inputs(img) == (160, 160, 3)
    assert ImageProcessing(img).eps == 1e-3
    with pytest.raises(ValueError):
        ImageProcessing(Config({"size": 160, "normalization": 7}))             # facenet.py:82


def test_flat_layout_and_counts_without_gpu():
    net = Network(embedding_size=128, device="cpu", allocate=False)
    assert net.count_variables() == (22808144, 22779312)
    assert len(net.layers) == 133 and net.CB == sum(L.cout for L in net.layers.values() if L.has_bn)
    offs = [(L.w_off, L.numel) for L in net.layers.values()]
    assert offs[0][0] == 0 and all(a + n == b for (a, n), (b, _) in zip(offs, offs[1:]))      # kernels are contiguous
    assert all(L.w_off % 8 == 0 and L.cin % 8 == 0 for L in net.layers.values())              # 16-B aligned packs
    assert net.n_decay >= net.n_kernel and net.beta_base == net.n_decay and net.n_params % 4 == 0
    first = net.layers["conv2d/Conv2d_1a_3x3"]
    assert (first.cin, first.cin_real) == (8, 3)
    # concat groups share one BN channel range in concat order (one BN pass per concat buffer)
    b = net.buf_bn["block35/0/mixed"]
    names = ["block35/0/tower_conv0/Conv2d_1x1", "block35/0/tower_conv1/Conv2d_0b_3x3", "block35/0/tower_conv2/Conv2d_0c_3x3"]
    assert [net.layers[n].bn_off - b for n in names] == [0, 32, 64]
    cls = Network(embedding_size=512, device="cpu", allocate=False, nrof_classes=10575)
    L = cls.layers["classifier/logits"]
    assert (L.cout, L.cout_real) == (10576, 10575)
    assert cls.count_variables()[0] == 23497424 + 512 * 10575 + 10575


def test_bucket_construction_covers_gradients_once():
    net = Network(embedding_size=128, device="cpu", allocate=False)
    layers = list(net.layers.values())
    # backward finishes layers roughly last-to-first: emulate with op index = reverse declaration order
    done_at = {L.index: 3 * (len(layers) - L.index) for L in layers}
    tail = (net.n_decay, net.n_params)
    for nb in (1, 4, 6, 16):
        b = parallel.make_buckets([L.w_off for L in layers], [L.numel for L in layers], done_at, net.n_kernel, tail, 3 * len(layers) + 5, nb)
        parallel.check_buckets(b, net.n_kernel, tail)
        assert len(b) <= nb + 2 and b[-1][1:] == tail
        assert b[0][2] == net.n_kernel and b[-2][1] == 0
    with pytest.raises(AssertionError):
        parallel.check_buckets([(0, 0, 10), (1, 20, 30)], 30, (30, 30))       # gap


def test_keras_variable_names_and_order():
    """The names and order Keras gives the reference's declaration (facenet/models/inception_resnet_v1.py:83-468): stage
    scopes, auto-numbered blocks / tower Sequentials / BatchNormalization layers, trainable variables first."""
    from facenet_amd import keras_names
    from facenet_amd.engine import Network
    net = Network(embedding_size=512, nrof_classes=10575, allocate=False)
    table = keras_names.keras_variable_table(net.layers)
    names = [k for k, _ in table]
    assert len(names) == len(set(names)) == 133 + 1 + 21 + 1 + 112 * 3        # kernels (+classifier), biases (+classifier), beta/mean/var
    assert names[0] == "inception_resnet_v1/conv2d/Conv2d_1a_3x3/kernel:0"
    assert names[1] == "inception_resnet_v1/conv2d/batch_normalization/beta:0"
    assert "inception_resnet_v1/block35/block35/sequential/Conv2d_1x1/kernel:0" in names          # first tower of the first Block35
    assert "inception_resnet_v1/block35/block35_4/sequential_14/Conv2d_0c_3x3/kernel:0" in names   # 15 tower Sequentials in block35
    assert "inception_resnet_v1/reduction_a/sequential_16/batch_normalization_39/beta:0" in names
    assert "inception_resnet_v1/block17/block17_9/sequential_36/Conv2d_0c_7x1/kernel:0" in names
    assert "inception_resnet_v1/reduction_b/sequential_39/batch_normalization_86/beta:0" in names
    assert "inception_resnet_v1/block8/block8_4/Conv2d_1x1/bias:0" in names                        # `up` of the fifth repeated Block8
    assert "inception_resnet_v1/block8_5/sequential_51/Conv2d_0c_3x1/kernel:0" in names            # the last Block8 (:453)
    assert "inception_resnet_v1/features/logits/kernel:0" in names
    assert "sequential_52/logits/bias:0" in names
    # model.weights: every trainable variable, then the moving statistics
    first_frozen = next(i for i, k in enumerate(names) if k.endswith("moving_mean:0"))
    assert all(not k.endswith(("moving_mean:0", "moving_variance:0")) for k in names[:first_frozen])
    assert all(k.endswith(("moving_mean:0", "moving_variance:0")) for k in names[first_frozen:])
    assert names[first_frozen] == "inception_resnet_v1/conv2d/batch_normalization/moving_mean:0"
    assert names[-1] == "inception_resnet_v1/features/batch_normalization_111/moving_variance:0"
    # name mapping is a bijection onto the engine's keys, and files are matched by name in either spelling
    params = {i: idx for idx, (_, i) in enumerate(table)}
    keras = keras_names.to_keras(params, net.layers)
    assert list(keras.keys()) == names
    back = keras_names.from_keras({k[:-2]: v for k, v in keras.items()}, net.layers)     # without the ':0' suffix
    assert back == params and keras_names.from_keras(params, net.layers) == params
    with pytest.raises(KeyError):
        keras_names.from_keras({k: v for k, v in keras.items() if "block17_3" not in k}, net.layers)
    assert keras_names.optimizer_slot_names(names[0]) == ("Adam/inception_resnet_v1/conv2d/Conv2d_1a_3x3/kernel/m:0",
                                                          "Adam/inception_resnet_v1/conv2d/Conv2d_1a_3x3/kernel/v:0")


def test_adam_iteration_count_survives_beta_power_underflow():
    """Advisor finding (round 2): Adam's t must be an integer of its own.  The fp32 running product beta1^t stops changing at the
    denormal floor (~970 steps at beta1 = 0.9), so t = log(power)/log(beta1) maps every later step to the same value; the
    powers are instead derived FROM t (host mirror of fn_adam_tick: pow in double, one rounding to fp32)."""
    import numpy as np
    from facenet_amd.train import adam_beta_powers
    run = np.float32(1.0)
    for _ in range(1100):
        run = np.float32(run * np.float32(0.9))
    recovered = 0 if run == 0 else int(round(float(np.log(run) / np.log(0.9))))
    assert recovered != 1100                                   # the defect: the product no longer identifies t
    b1, b2 = adam_beta_powers(1100, 0.9, 0.999)
    assert b1 == 0.0                                           # 0.9^1100 = 4.6e-51 underflows to +0 like Keras' fp32 pow
    assert abs(b2 - float(np.float32(0.999)) ** 1100) < 1e-7   # the betas are the fp32 values the device holds
    assert adam_beta_powers(0, 0.9, 0.999) == (1.0, 1.0)
    assert adam_beta_powers(1, 0.9, 0.999) == (float(np.float32(0.9)), float(np.float32(0.999)))
    for t in (10, 500, 969, 980, 1000):                        # distinct t stay distinct through beta2^t long after beta1^t is gone
        assert adam_beta_powers(t, 0.9, 0.999)[1] > adam_beta_powers(t + 1, 0.9, 0.999)[1]
