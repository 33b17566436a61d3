"""Keras variable names and order of the reference network.

The reference saves and restores weights through Keras (``network.load_weights`` / ``ModelCheckpoint(save_weights_only=True)``,
apps/train_softmax.py:68-78,105).  What Keras writes is keyed by the variables its layer declaration creates, so this module
derives those names -- and the order of ``model.weights`` -- from the declaration in facenet/models/inception_resnet_v1.py:

* the model is ``InceptionResnetV1`` -> scope ``inception_resnet_v1``; its stages are the named Sequentials ``conv2d``,
  ``block35``, ``block17``, ``block8``, ``features`` (:387,435,443,451,459) and the auto-named layers ``reduction_a``,
  ``reduction_b`` (:438,446) and the last ``Block8`` (:453);
* un-named layers get ``snake_case(class)`` + ``_N`` from a per-class counter in creation order: the residual blocks are
  ``block35`` .. ``block35_4``, ``block17`` .. ``block17_9``, ``block8`` .. ``block8_5``; every tower is an un-named
  ``Sequential`` (``sequential`` .. ``sequential_51``); the 112 ``BatchNormalization`` layers are ``batch_normalization`` ..
  ``batch_normalization_111`` (:56-63 passes no name);
* Conv2D / Dense layers are named in the declaration (``Conv2d_1a_3x3`` .., ``logits``); variables are ``kernel``, ``bias``,
  ``beta``, ``moving_mean``, ``moving_variance`` (``scale=False``: no gamma);
* ``model.weights`` lists every trainable variable in declaration order first, then the non-trainable ones
  (``moving_mean``, ``moving_variance`` per BatchNormalization, same order).

Parity unpinned: TensorFlow is not installed here and the reference ships no checkpoint, so the naming follows the
documented TF-2.4 Keras rules rather than a file.  ``Network.load_keras_params`` therefore also accepts the engine's own
``<layer>/kernel`` keys (``internal_keys``), and files are matched by name, never by position.
"""
from __future__ import annotations

import re
from typing import Dict, List, Tuple

MODEL_SCOPE = "inception_resnet_v1"
TRAIN_SCOPE = "sequential_52"       # apps/train_softmax.py:57: the un-named Sequential([model, Dense]) is the 53rd one created


def _numbered(base: str, n: int) -> str:
    return base if n == 0 else f"{base}_{n}"


def _scope_of(layer_name: str, seq_counter: Dict[str, int], seq_ids: Dict[str, int]) -> Tuple[str, str]:
    """internal layer name -> (Keras scope path of the layer's parent, Keras layer name)."""
    parts = layer_name.split("/")
    leaf = parts[-1]
    if parts[0] == "conv2d":
        return f"{MODEL_SCOPE}/conv2d", leaf
    if parts[0] == "features":
        return f"{MODEL_SCOPE}/features", leaf
    if parts[0] == "classifier":
        return TRAIN_SCOPE, leaf
    m = re.fullmatch(r"(block35|block17|block8)", parts[0])
    if m:                                   # "block17/3/tower_conv1/Conv2d_0b_1x7" or "block17/3/up"
        stage, idx = parts[0], int(parts[1])
        block = f"{MODEL_SCOPE}/{stage}/{_numbered(stage, idx)}"
        rest = parts[2:]
    elif parts[0] == "block8_2":            # the last Block8 (:453) is a direct child of the model: block8_5 after five repeats
        block = f"{MODEL_SCOPE}/{_numbered('block8', seq_counter['block8_repeat'])}"
        rest = parts[1:]
    else:                                   # reduction_a / reduction_b
        block = f"{MODEL_SCOPE}/{parts[0]}"
        rest = parts[1:]
    if rest[0] == "up":
        return block, "Conv2d_1x1"
    tower_key = "/".join(parts[:-1])
    if tower_key not in seq_ids:
        seq_ids[tower_key] = seq_counter["sequential"]
        seq_counter["sequential"] += 1
    return f"{block}/{_numbered('sequential', seq_ids[tower_key])}", leaf


def keras_variable_table(layers, block8_repeat: int = 5) -> List[Tuple[str, str]]:
    """[(keras variable name, internal key)] in ``model.weights`` order for an ordered mapping of engine layers
    (``Network.layers``: declaration order, attributes name / has_bias / has_bn)."""
    seq_counter = {"sequential": 0, "block8_repeat": block8_repeat}
    seq_ids: Dict[str, int] = {}
    bn = 0
    trainable: List[Tuple[str, str]] = []
    frozen: List[Tuple[str, str]] = []
    # Keras tracks a block's sub-layers in attribute order: the towers first, ``up`` last -- the engine declares them the same way
    for L in layers.values():
        scope, name = _scope_of(L.name, seq_counter, seq_ids)
        trainable.append((f"{scope}/{name}/kernel:0", L.name + "/kernel"))
        if L.has_bias:
            trainable.append((f"{scope}/{name}/bias:0", L.name + "/bias"))
        if L.has_bn:
            pre = "features/bn" if L.name == "features/logits" else L.name + "/bn"
            b = f"{scope}/{_numbered('batch_normalization', bn)}"
            bn += 1
            trainable.append((f"{b}/beta:0", pre + "/beta"))
            frozen.append((f"{b}/moving_mean:0", pre + "/moving_mean"))
            frozen.append((f"{b}/moving_variance:0", pre + "/moving_variance"))
    return trainable + frozen


def to_keras(params: Dict, layers, block8_repeat: int = 5) -> "Dict":
    """engine keys -> Keras variable names, in ``model.weights`` order."""
    from collections import OrderedDict
    return OrderedDict((k, params[i]) for k, i in keras_variable_table(layers, block8_repeat) if i in params)


def from_keras(params: Dict, layers, block8_repeat: int = 5) -> "Dict":
    """Keras variable names (with or without the ':0' suffix) or engine keys -> engine keys."""
    out = {}
    table = keras_variable_table(layers, block8_repeat)
    for k, i in table:
        for cand in (k, k[:-2], i):
            if cand in params:
                out[i] = params[cand]
                break
    missing = [k for k, i in table if i not in out]
    if missing:
        raise KeyError(f"weights file lacks {len(missing)} variables, first: {missing[0]}")   # h5utils.py:65 raises KeyError too
    return out


def optimizer_slot_names(var_name: str) -> Tuple[str, str]:
    """Keras Adam slot variables of one model variable: 'Adam/<var>/m:0', 'Adam/<var>/v:0'."""
    base = var_name[:-2] if var_name.endswith(":0") else var_name
    return f"Adam/{base}/m:0", f"Adam/{base}/v:0"
