"""Hot-path configuration.

The shims in this package are duck-typed on ``config``: the reference's own ``Config`` object
(`utils/config.py:14-408`, loaded by `Config.load`, `utils/config.py:410-910`) works unchanged.
``HotPathConfig`` is a minimal stand-in that carries only the attributes the SDF-training hot path
reads (SURVEY.md section 8 header), with the reference's defaults, and reads the same YAML
sections/keys so `config/run_ncd128.yaml` / `run_SubT_MRS.yaml` resolve to the same values.
"""
from __future__ import annotations

import os

import torch
import yaml


class HotPathConfig:
    def __init__(self) -> None:
        # setting
        self.name = "clid_native"
        self.device = "cuda"
        self.dtype = torch.float32
        self.silence = True
        self.seed = 42
        self.semantic_on = False
        self.color_on = False
        self.color_channel = 0
        # process
        self.min_range = 1.0
        self.max_range = 60.0
        self.vox_down_m = 0.1
        # sampler (`utils/config.py:518-540`)
        self.surface_sample_range_m = 0.25
        self.surface_sample_n = 4
        self.free_sample_begin_ratio = 0.5
        self.free_sample_end_dist_m = 1.2
        self.free_front_n = 2
        self.free_behind_n = 1
        # raw-point map of the region-specific SDF labels (`utils/config.py:111,124,148,518-521`)
        self.use_pin_mapper = False
        self.local_voxel_size_m = 0.2
        self.local_buffer_size = int(5e6)
        self.local_map_size = 100.0
        # per-frame glue of Mapper.process_frame (`utils/config.py:105-107,132-144,162-165,238-241`)
        self.from_sample_points = True
        self.from_all_samples = False
        self.map_surface_ratio = 0.5
        self.prune_map_on = False
        self.prune_freq_frame = 100
        self.max_prune_certainty = 3.0
        self.pool_filter_freq = 1
        self.new_certainty_thre = 1.0
        self.new_sample_ratio_less = 0.02
        self.new_sample_ratio_more = 0.15
        self.new_sample_ratio_restart = 0.3
        self.dynamic_filter_on = False
        self.dynamic_certainty_thre = 0.5
        self.dynamic_sdf_ratio_thre = 1.5
        self.dynamic_min_grad_norm_thre = 0.3
        self.track_on = False
        self.pgo_on = False
        # neural points (`utils/config.py:542-600`)
        self.voxel_size_m = 0.4
        self.buffer_size = int(5e7)
        self.feature_dim = 8
        self.feature_std = 0.0
        self.query_nn_k = 6
        self.num_nei_cells = 2
        self.search_alpha = 0.5
        self.weighted_first = True
        self.layer_norm_on = False
        self.pos_encoding_band = 0
        self.pos_input_dim = 3
        self.use_gaussian_pe = False
        self.use_mid_ts = False
        self.local_map_travel_dist_ratio = 5.0
        self.diff_ts_local = 400.0
        # decoder
        self.geo_mlp_level = 1
        self.geo_mlp_hidden_dim = 64
        self.mlp_bias_on = True
        self.mlp_leaky_relu = False
        self.freeze_after_frame = 40
        # loss (`utils/config.py:622-657`)
        self.main_loss_type = "bce"
        self.sigma_sigmoid_m = 0.1
        self.logistic_gaussian_ratio = 0.55
        self.loss_weight_on = True
        self.dist_weight_on = True
        self.dist_weight_scale = 0.8
        self.behind_dropoff_on = False
        self.ekional_loss_on = True
        self.ekional_add_to = "all"
        self.weight_e = 0.5
        self.numerical_grad = True
        self.gradient_decimation = 10
        self.num_grad_step_ratio = 0.2
        self.proj_correction_on = False
        self.consistency_loss_on = False
        self.weight_c = 0.5            # `utils/config.py:215-219`
        self.consistency_count = 1000  # (derived from bs below, `utils/config.py:904`)
        self.consistency_range = 0.05
        # continual
        self.bs_new_sample = 1000
        self.pool_capacity = int(1e7)
        # optimizer (`utils/config.py:800-835`)
        self.iters = 10
        self.init_iter_ratio = 40
        self.bs = 16384
        self.lr = 0.01
        self.lr_pose = 1e-4
        self.weight_decay = 0.0
        self.adam_eps = 1e-15
        self.opt_adam = True
        self.adaptive_iters = True
        self.ba_freq_frame = 0
        self.wandb_vis_on = False
        # tracker keys read by the measurement model (`utils/config.py:264-274`)
        self.tran_dtype = torch.float64
        self.reg_min_grad_norm = 0.5
        self.reg_max_grad_norm = 1.5
        self.track_mask_query_nn_k = self.query_nn_k
        self.max_sdf_std_ratio = 1.0
        self.reg_iter_n = 50
        self._derive()

    def _derive(self) -> None:
        # `utils/config.py:903-910`
        self.infer_bs = self.bs * 64
        self.consistency_count = int(self.bs / 4)
        self.local_map_radius = self.max_range + 2.0
        self.window_radius = max(self.max_range, 6.0)

    def load(self, config_file: str) -> "HotPathConfig":
        with open(os.path.abspath(config_file)) as fh:
            args = yaml.safe_load(fh) or {}
        g = args.get
        s = g("setting", {})
        self.name = s.get("name", self.name)
        self.device = s.get("device", self.device)
        self.seed = s.get("random_seed", self.seed)
        self.use_pin_mapper = s.get("use_pin_mapper", self.use_pin_mapper)
        p = g("process", {})
        self.min_range = p.get("min_range_m", self.min_range)
        self.max_range = p.get("max_range_m", self.max_range)
        self.vox_down_m = p.get("vox_down_m", self.vox_down_m)
        if "sampler" in args:  # `utils/config.py:518-540`: absent keys fall back to values DERIVED from vox_down_m
            sa = args["sampler"] or {}
            self.local_voxel_size_m = sa.get("local_voxel_size_m", self.vox_down_m)
            self.surface_sample_range_m = sa.get("surface_sample_range_m", self.vox_down_m * 3.0)
            self.free_sample_begin_ratio = sa.get("free_sample_begin_ratio", self.free_sample_begin_ratio)
            self.free_sample_end_dist_m = sa.get("free_sample_end_dist_m", self.surface_sample_range_m * 4.0)
            self.surface_sample_n = sa.get("surface_sample_n", self.surface_sample_n)
            self.free_front_n = sa.get("free_front_sample_n", self.free_front_n)
            self.free_behind_n = sa.get("free_behind_sample_n", self.free_behind_n)
        n = g("neuralpoints", {}) or {}
        if "neuralpoints" in args:  # `utils/config.py:542-545`
            self.voxel_size_m = n.get("voxel_size_m", self.vox_down_m * 5.0)
        self.query_nn_k = n.get("query_nn_k", self.query_nn_k)
        self.buffer_size = int(float(n.get("buffer_size", self.buffer_size)))
        self.num_nei_cells = n.get("num_nei_cells", self.num_nei_cells)
        self.layer_norm_on = n.get("layer_norm_on", self.layer_norm_on)
        self.search_alpha = n.get("search_alpha", self.search_alpha)
        self.feature_dim = n.get("feature_dim", self.feature_dim)
        self.feature_std = n.get("feature_std", self.feature_std)
        self.weighted_first = n.get("weighted_first", self.weighted_first)
        self.use_mid_ts = n.get("use_mid_ts", self.use_mid_ts)
        self.local_map_travel_dist_ratio = n.get(
            "local_map_travel_dist_ratio", self.local_map_travel_dist_ratio
        )
        d = g("decoder", {})
        self.geo_mlp_level = d.get("mlp_level", self.geo_mlp_level)
        self.geo_mlp_hidden_dim = d.get("mlp_hidden_dim", self.geo_mlp_hidden_dim)
        self.freeze_after_frame = d.get("freeze_after_frame", self.freeze_after_frame)
        lo = g("loss", {}) or {}
        self.main_loss_type = lo.get("main_loss_type", self.main_loss_type)
        if "loss" in args:  # `utils/config.py:624-626`
            self.sigma_sigmoid_m = lo.get("sigma_sigmoid_m", self.vox_down_m)
        self.loss_weight_on = lo.get("loss_weight_on", self.loss_weight_on)
        if self.loss_weight_on:  # `utils/config.py:630-637`
            self.dist_weight_scale = lo.get("dist_weight_scale", self.dist_weight_scale)
            self.behind_dropoff_on = lo.get("behind_dropoff_on", self.behind_dropoff_on)
        self.ekional_loss_on = lo.get("ekional_loss_on", self.ekional_loss_on)
        self.weight_e = float(lo.get("weight_e", self.weight_e))
        self.numerical_grad = lo.get("numerical_grad_on", self.numerical_grad)
        if not self.numerical_grad:  # `utils/config.py:645-646`
            self.gradient_decimation = 1
        else:
            self.gradient_decimation = lo.get("grad_decimation", self.gradient_decimation)
            self.num_grad_step_ratio = lo.get("num_grad_step_ratio", self.num_grad_step_ratio)
        self.consistency_loss_on = lo.get("consistency_loss_on", self.consistency_loss_on)  # `utils/config.py:654-656`
        self.from_sample_points = n.get("from_sample_points", self.from_sample_points)
        self.map_surface_ratio = n.get("map_surface_ratio", self.map_surface_ratio)
        self.prune_map_on = n.get("prune_map_on", self.prune_map_on)
        self.max_prune_certainty = n.get("max_prune_certainty", self.max_prune_certainty)
        self.dynamic_filter_on = p.get("dynamic_filter_on", self.dynamic_filter_on)
        self.dynamic_sdf_ratio_thre = p.get("dynamic_sdf_ratio_thre", self.dynamic_sdf_ratio_thre)
        self.dynamic_certainty_thre = p.get("dynamic_certainty_thre", self.dynamic_certainty_thre)
        c = g("continual", {})
        self.new_certainty_thre = float(c.get("new_certainty_thre", self.new_certainty_thre))
        self.pool_filter_freq = c.get("pool_filter_freq", self.pool_filter_freq)
        self.bs_new_sample = int(c.get("batch_size_new_sample", self.bs_new_sample))
        self.pool_capacity = int(float(c.get("pool_capacity", self.pool_capacity)))
        # `utils/config.py:676, 742-743`: tracking is on only when the YAML has a `tracker` section, PGO only with
        # tracking and a `pgo` section -- determine_used_pose picks odom / pgo / gt poses from these
        self.track_on = bool(g("tracker", False))
        self.pgo_on = bool(g("pgo", False)) if self.track_on else False
        if self.track_on:  # the measurement model's keys (`utils/config.py:718-727`)
            tr = args["tracker"]
            self.reg_iter_n = tr.get("iter_n", self.reg_iter_n)
            self.track_mask_query_nn_k = tr.get("valid_nn_k", self.track_mask_query_nn_k)
            self.reg_min_grad_norm = tr.get("min_grad_norm", self.reg_min_grad_norm)
            self.reg_max_grad_norm = tr.get("max_grad_norm", self.reg_max_grad_norm)
        o = g("optimizer", {})
        self.iters = o.get("iters", self.iters)
        self.bs = o.get("batch_size", self.bs)
        self.lr = float(o.get("learning_rate", self.lr))
        self.adaptive_iters = o.get("adaptive_iters", self.adaptive_iters)
        self.new_sample_ratio_less = o.get("new_sample_ratio_less", self.new_sample_ratio_less)
        self.new_sample_ratio_more = o.get("new_sample_ratio_more", self.new_sample_ratio_more)
        self._derive()
        return self
