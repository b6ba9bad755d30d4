"""Stand-in modules for the reference's UN-VENDORED imports, so that the reference's own
``src/slam/coslam/model/{scene_rep,decoder}.py`` and ``coslam_utils.py`` can be imported verbatim
in the build container.

TEST INFRASTRUCTURE ONLY (used by oracle/make_golden.py).  Everything here is this repo's own
restatement of the published algorithms of

* ``tinycudann`` (NVlabs/tiny-cuda-nn, unpinned HEAD; reference README.md:171-173), and
* ``third_parties.coslam`` (HengyiWang/Co-SLAM @ 3bb904e; reference .gitmodules:1-3,
  scripts/installation/conda_env/build.sh:22-23),

both absent from /root/reference -- i.e. the "parity unpinned" rows.  The arithmetic lives in
oracle/spec_torch.py; this file only gives it the module / class names the reference imports.
"""

from __future__ import annotations

import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import spec_torch as S


# ---------------------------------------------------------------- tinycudann
class _Encoding(nn.Module):
    def __init__(self, n_input_dims, encoding_config, dtype=None):
        super().__init__()
        self.otype = encoding_config["otype"]
        self.n_input_dims = n_input_dims
        if self.otype == "HashGrid":
            self.meta = S.HashGridMeta(
                encoding_config["n_levels"], encoding_config["n_features_per_level"],
                encoding_config["log2_hashmap_size"], encoding_config["base_resolution"],
                encoding_config["per_level_scale"])
            self.params = nn.Parameter((torch.rand(self.meta.n_params) * 2 - 1) * 1e-4)
            self.n_output_dims = self.meta.n_output_dims
        elif self.otype == "OneBlob":
            self.n_bins = encoding_config["n_bins"]
            self.params = nn.Parameter(torch.zeros(0))
            self.n_output_dims = n_input_dims * self.n_bins
        else:
            raise NotImplementedError(self.otype)

    def forward(self, x):
        if self.otype == "HashGrid":
            return S.hash_encode(x, self.params, self.meta)
        return S.oneblob_encode(x, self.n_bins)


# ---------------------------------------------------------------- third_parties.coslam.model.encodings
def get_encoder(encoding, input_dim=3, degree=4, n_bins=16, n_frequencies=12, n_levels=16, level_dim=2,
                base_resolution=16, log2_hashmap_size=19, desired_resolution=512):
    if 'hash' in encoding.lower() or 'tiled' in encoding.lower():
        per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (n_levels - 1))
        embed = _Encoding(input_dim, {"otype": 'HashGrid', "n_levels": n_levels,
                                      "n_features_per_level": level_dim,
                                      "log2_hashmap_size": log2_hashmap_size,
                                      "base_resolution": base_resolution,
                                      "per_level_scale": per_level_scale}, dtype=torch.float)
    elif 'blob' in encoding.lower():
        embed = _Encoding(input_dim, {"otype": "OneBlob", "n_bins": n_bins}, dtype=torch.float)
    else:
        raise NotImplementedError(encoding)
    return embed, embed.n_output_dims


# ---------------------------------------------------------------- third_parties.coslam.model.utils
def batchify(fn, chunk=1024 * 64):
    if chunk is None:
        return fn
    raise NotImplementedError


def compute_loss(prediction, target, loss_type='l2'):
    if loss_type == 'l2':
        return F.mse_loss(prediction, target)
    if loss_type == 'l1':
        return F.l1_loss(prediction, target)
    raise Exception('Unsupported loss type')


def get_sdf_loss(z_vals, target_d, predicted_sdf, truncation, loss_type=None, grad=None):
    return S.get_sdf_loss(z_vals, target_d, predicted_sdf, truncation)


def sample_pdf(*a, **k):
    raise NotImplementedError("n_importance is 0 in every shipped config")


# ---------------------------------------------------------------- third_parties.coslam.model.decoder
class SDFNet(nn.Module):
    pass


class ColorSDFNet(nn.Module):
    pass


class ColorNet(nn.Module):
    def __init__(self, config, input_ch=4, geo_feat_dim=15, hidden_dim_color=64, num_layers_color=3):
        super().__init__()
        layers = []
        for l in range(num_layers_color):
            in_dim = input_ch + geo_feat_dim if l == 0 else hidden_dim_color
            out_dim = 3 if l == num_layers_color - 1 else hidden_dim_color
            layers.append(nn.Linear(in_dim, out_dim, bias=False))
            if l != num_layers_color - 1:
                layers.append(nn.ReLU(inplace=True))
        self.model = nn.Sequential(*layers)

    def forward(self, input_feat):
        return self.model(input_feat)


# ---------------------------------------------------------------- third_parties.coslam.model.scene_rep
class JointEncoding(nn.Module):
    def get_resolution(self):
        self.resolution_sdf = S.get_resolution(self.bounding_box, self.config['grid']['voxel_sdf'])
        self.resolution_color = S.get_resolution(self.bounding_box, self.config['grid']['voxel_color'])

    def get_encoding(self, config):
        self.embedpos_fn, self.input_ch_pos = get_encoder(config['pos']['enc'], n_bins=self.config['pos']['n_bins'])
        self.embed_fn, self.input_ch = get_encoder(config['grid']['enc'],
                                                   log2_hashmap_size=config['grid']['hash_size'],
                                                   desired_resolution=self.resolution_sdf)

    def sdf2weights(self, sdf, z_vals, args=None):
        return S.sdf2weights(sdf, z_vals, args['training']['trunc'], args['data']['sc_factor'])

    def query_color(self, query_points):
        return torch.sigmoid(self.query_color_sdf(query_points)[..., :3])

    def run_network(self, inputs):
        inputs_flat = torch.reshape(inputs, [-1, inputs.shape[-1]])
        if self.config['grid']['tcnn_encoding']:
            inputs_flat = (inputs_flat - self.bounding_box[:, 0]) / (self.bounding_box[:, 1] - self.bounding_box[:, 0])
        outputs_flat = batchify(self.query_color_sdf, None)(inputs_flat)
        return torch.reshape(outputs_flat, list(inputs.shape[:-1]) + [outputs_flat.shape[-1]])


# ---------------------------------------------------------------- third_parties.coslam.utils
def getVoxels(x_max, x_min, y_max, y_min, z_max, z_min, voxel_size=None, resolution=None):
    bb = torch.tensor([[float(x_min), float(x_max)], [float(y_min), float(y_max)], [float(z_min), float(z_max)]],
                      dtype=torch.float64)
    return S.get_voxels(bb, voxel_size)


def get_batch_query_fn(query_fn, num_args=1, device=None):
    if num_args == 1:
        return lambda f, i0, i1: query_fn(f[i0:i1, None, :].to(device))
    return lambda f, f1, i0, i1: query_fn(f[i0:i1, None, :].to(device), f1[i0:i1, :].to(device))


def install():
    """Register the stand-ins under the names the reference imports."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("mmengine", Config=dict)
    mod("tinycudann", Encoding=_Encoding, Network=None)
    mod("marching_cubes")
    mod("trimesh", Trimesh=object)
    if "matplotlib" not in sys.modules:
        try:
            import matplotlib.pyplot  # noqa: F401
        except Exception:
            mod("matplotlib", pyplot=mod("matplotlib.pyplot"))
    tp = mod("third_parties")
    cs = mod("third_parties.coslam")
    md = mod("third_parties.coslam.model")
    tp.coslam = cs
    cs.model = md
    md.scene_rep = mod("third_parties.coslam.model.scene_rep", JointEncoding=JointEncoding)
    md.decoder = mod("third_parties.coslam.model.decoder", SDFNet=SDFNet, ColorNet=ColorNet, ColorSDFNet=ColorSDFNet)
    md.utils = mod("third_parties.coslam.model.utils", sample_pdf=sample_pdf, get_sdf_loss=get_sdf_loss,
                   mse2psnr=S.mse2psnr, compute_loss=compute_loss, batchify=batchify)
    md.encodings = mod("third_parties.coslam.model.encodings", get_encoder=get_encoder)
    cs.utils = mod("third_parties.coslam.utils", getVoxels=getVoxels, get_batch_query_fn=get_batch_query_fn)
