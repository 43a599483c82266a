"""Build the product's shim objects (clid_slam_amd.NeuralPoints / Decoder / Mapper) from the golden
state so the HIP path and the oracle see identical inputs."""
import numpy as np
import torch

import golden_io as gio
from clid_slam_amd import Decoder, HotPathConfig, Mapper, NeuralPoints


class DatasetStub:
    lose_track = False
    stop_status = False
    processed_frame = 2
    gt_pose_provided = False


def config(device="cuda", **over):
    z = gio.load("state.npz")
    cfg = HotPathConfig()
    cfg.device = device
    cfg.buffer_size = int(gio.S(z["buffer_size"]))
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def neural_points(cfg, z=None, base=None):
    """NeuralPoints shim holding exactly the golden map state (optionally the pre-loop `base`
    features/certainties of pool.npz)."""
    z = z or gio.load("state.npz")
    dev = cfg.device
    nm = NeuralPoints(cfg)
    B = cfg.buffer_size
    tab = torch.full((B,), -1, dtype=torch.int64)
    tab[gio.T(z["table_slot"])] = gio.T(z["table_idx"])
    nm.buffer_pt_index = tab.to(dev)
    nm.neural_points = gio.T(z["neural_points"]).to(dev)
    nm.point_orientations = torch.zeros((nm.neural_points.shape[0], 4), device=dev)
    nm.point_ts_create = gio.T(z["point_ts_create"]).to(dev)
    nm.point_ts_update = gio.T(z["point_ts_update"]).to(dev)
    nm.travel_dist = gio.T(z["travel_dist"]).to(dev)
    nm.cur_ts = int(gio.S(z["cur_ts"]))
    nm.global2local = gio.T(z["global2local"]).to(dev)
    nm.local_mask = gio.T(z["local_mask"]).to(dev)
    nm.local_neural_points = gio.T(z["local_neural_points"]).to(dev)
    nm.local_point_orientations = torch.zeros((nm.local_neural_points.shape[0], 4), device=dev)
    geo = z["geo_features"] if base is None else base["base_geo_features"]
    cert = z["point_certainties"] if base is None else base["base_point_certainties"]
    nm.geo_features = gio.T(geo).clone().to(dev)
    nm.point_certainties = gio.T(cert).clone().to(dev)
    if base is not None:
        nm.point_ts_update = gio.T(base["base_point_ts_update"]).clone().to(dev)
    m = nm.local_mask
    nm.local_geo_features = torch.nn.Parameter(nm.geo_features[m].clone())
    nm.local_point_certainties = nm.point_certainties[m[:-1]].clone()
    nm.local_point_ts_update = nm.point_ts_update[m[:-1]].clone()
    nm.local_map_radius = 12.0
    nm._local_ids = torch.nonzero(m[:-1]).flatten().contiguous()
    nm._map_version += 1
    return nm


def decoder(cfg, z=None, prefix=""):
    z = z or gio.load("state.npz")
    dec = Decoder(cfg, cfg.geo_mlp_hidden_dim, cfg.geo_mlp_level, 1)
    with torch.no_grad():
        dec.layers[0].weight.copy_(gio.T(z[prefix + "W1"]))
        dec.layers[0].bias.copy_(gio.T(z[prefix + "b1"]))
        dec.lout.weight.copy_(gio.T(z[prefix + "W2"]))
        dec.lout.bias.copy_(gio.T(z[prefix + "b2"]))
    return dec


def mapper(cfg, nm, dec, new_idx=None):
    p = gio.load("pool.npz")
    mp = Mapper(cfg, DatasetStub(), nm, None, dec)
    mp.set_pool(gio.T(p["coord"]), gio.T(p["sdf_label"]), gio.T(p["weight"]), gio.T(p["time"]), new_idx)
    return mp, p


def task_records(rec, n_it, n_tasks):
    """[n_it, n_tasks, 48, 4] view of the task records clid_train_search wrote (an iteration's block = its n_tasks x 192
    floats of records followed by its tile number blocks, include/clid_native.h clid_train_search_tasks)."""
    per = rec.numel() // n_it
    return rec.view(n_it, per)[:, : n_tasks * 192].reshape(n_it, n_tasks, 48, 4)
