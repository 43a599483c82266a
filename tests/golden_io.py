"""Load the committed golden fixtures (tests/golden/*.npz, produced by oracle/make_golden.py from
the reference itself) into the oracle's MapState / DecoderParams / SamplePool."""
import os

import numpy as np
import torch

from oracle import cpu_ref as O

# (CLID_GOLDEN_DIR: another directory of reference-generated fixtures -- `oracle/make_golden.py --fresh DIR --seed N`; the
# out-of-fixture run of tests/test_oracle_golden.py)
GOLDEN = os.environ.get("CLID_GOLDEN_DIR") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def S(a):
    return np.asarray(a).reshape(-1)[0].item()


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def map_state(z=None, layer_norm_on=False, weighted_first=True):
    z = z or load("state.npz")
    B = int(S(z["buffer_size"]))
    table = torch.full((B,), -1, dtype=torch.int64)
    table[T(z["table_slot"])] = T(z["table_idx"])
    st = O.MapState(
        buffer_pt_index=table,
        neural_points=T(z["neural_points"]),
        point_ts_create=T(z["point_ts_create"]),
        travel_dist=T(z["travel_dist"]),
        cur_ts=int(S(z["cur_ts"])),
        global2local=T(z["global2local"]),
        local_neural_points=T(z["local_neural_points"]),
        local_geo_features=T(z["local_geo_features"]).clone(),
        local_point_certainties=T(z["local_point_certainties"]).clone(),
        local_point_ts_update=T(z["local_point_ts_update"]).clone(),
        geo_features=T(z["geo_features"]).clone(),
        point_certainties=T(z["point_certainties"]).clone(),
        resolution=float(S(z["resolution"])),
        buffer_size=B,
        diff_travel_dist_local=float(S(z["diff_travel_dist_local"])),
        neighbor_dx=T(z["neighbor_dx"]),
        max_valid_dist2=float(S(z["max_valid_dist2"])),
        layer_norm_on=layer_norm_on,
        weighted_first=weighted_first,
    )
    return st


def decoder(z=None, prefix=""):
    z = z or load("state.npz")
    return O.DecoderParams(
        T(z[prefix + "W1"]).clone(), T(z[prefix + "b1"]).clone(), T(z[prefix + "W2"]).clone(),
        T(z[prefix + "b2"]).clone(), float(S(load("state.npz")["sdf_scale"])),
    )


def sample_pool():
    p = load("pool.npz")
    return O.SamplePool(T(p["coord"]), T(p["sdf_label"]), T(p["time"]), T(p["weight"])), p


def sampler_noise(seed, n_rays, n_surface=4, n_front=2, n_behind=1):
    """The random draws of DataSampler.sample / sample_pin for one frame, in the reference's order from
    torch's CPU generator seeded with `seed` (utils/data_sampler.py:47, :72, :93)."""
    g = torch.Generator().manual_seed(int(seed))
    z_s = torch.randn(n_rays * n_surface, 1, generator=g)
    u_f = torch.rand(n_rays * n_front, 1, generator=g)
    u_b = torch.rand(n_rays * n_behind, 1, generator=g)
    return z_s, u_f, u_b


def as_double(obj):
    """A copy-free conversion of a dataclass' float32 tensors to float64 (the oracle in double precision: the arbiter where two
    fp32 summation orders of a 400 k-term sum disagree)."""
    import dataclasses

    for f in dataclasses.fields(obj):
        v = getattr(obj, f.name)
        if isinstance(v, torch.Tensor) and v.dtype == torch.float32:
            setattr(obj, f.name, v.double())
    return obj
