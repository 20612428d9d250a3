"""CLI of the training / evaluation drivers: every flag of ``/root/reference/utils/arguments.py:29-313`` with
the same name, type, default and list-collapsing rule (``:19-24``: a list argument of length 1 becomes a scalar,
longer lists become tuples), declared as a table; plus a few flags of this build (synthetic data, compute
dtype, hipGraph replay).  Flags whose subsystem is out of the hot-path scope (learned class vectors, MONAI
transforms, Neptune) are accepted and ignored so reference command lines keep working."""
from __future__ import annotations

import argparse

S, I, F = str, int, float
# (flag, kind, default, type)   kind: v = value, l = list (nargs='*'), t = store_true, f = store_false(dest)
_SPEC = {
    "model": [
        ("--model", "v", "UNETR_Official", S), ("--vol_size", "l", [96], I), ("--patch_size", "l", [16], I),
        ("--window_size", "l", [6], I), ("--input_dim", "v", 3, I), ("--output_dim", "v", 3, I),
        ("--in_chans", "v", 1, I), ("--hidden_dim", "v", 48, I), ("--depths", "l", [2, 2, 2, 2], I),
        ("--num_heads", "l", [3, 6, 12, 24], I), ("--mlp_ratio", "v", 4.0, F), ("--rel_pos_bias", "t"),
        ("--rel_pos_bias_affine", "t"), ("--abs_pos_emb", "t"), ("--rel_crop_pos_emb", "t"), ("--qkv_bias", "t"),
        ("--gradient_clipping", "v", None, F), ("--mixed_precision", "t"), ("--learned_cls_vectors", "t"),
        ("--lcv_vector_dim", "v", 6, I), ("--lcv_final_layer", "t"), ("--lcv_sincos_emb", "t"),
        ("--lcv_concat_vector", "t"), ("--lcv_only", "t"), ("--lcv_linear_comb", "t"), ("--lcv_patch_voxel_mean", "t"),
        ("--use_abs_pos_emb", "t"), ("--global_token", "t"),
    ],
    "transform": [
        ("--t_voxel_spacings", "t"), ("--t_voxel_dims", "l", [1.0], F), ("--t_cubed_ct_intensity", "t"),
        ("--t_fixed_ct_intensity", "t"), ("--t_percentile_ct_intensity", "t"), ("--t_ct_min", "v", -1000, I),
        ("--t_ct_max", "v", 1000, I), ("--t_crop_foreground_img", "t"), ("--t_crop_foreground_kdiv", "t"),
        ("--t_rand_crop_fgbg", "t"), ("--t_rand_crop_pos_weight", "v", 1.0, F), ("--t_rand_crop_neg_weight", "v", 1.0, F),
        ("--t_rand_crop_classes", "t"), ("--t_rand_crop_dilated_center", "t"), ("--t_rand_spatial_crop", "t"),
        ("--t_spatial_pad", "t"), ("--t_convert_labels_to_brats", "t"), ("--t_normalize", "t"),
        ("--t_normalize_channel_wise", "t"), ("--t_norm_mean", "v", 0.1943, F), ("--t_norm_std", "v", 0.2786, F),
        ("--t_n_patches_per_image", "v", 1, I), ("--t_flip_prob", "v", 0.0, F), ("--t_rot_prob", "v", 0.0, F),
        ("--t_intensity_shift_os", "v", 0.1, F), ("--t_intensity_shift_prob", "v", 0.0, F),
        ("--t_intensity_scale_factors", "v", 0.1, F), ("--t_intensity_scale_prob", "v", 0.0, F),
    ],
    "data": [
        ("--data_path", "v", "/datasets/", S), ("--json_list", "v", "dataset.json", S), ("--task", "v", "Task03_Liver", S),
        ("--batch_size_val", "v", 1, I), ("--n_images_per_batch", "v", 8, I), ("--n_workers_train", "v", 8, I),
        ("--n_workers_val", "v", 2, I), ("--no_pin_memory", "f", "pin_mem"), ("--no_cache_dataset", "f", "cache_dataset"),
        ("--cache_rate_train", "v", 1.0, F), ("--cache_rate_val", "v", 1.0, F),
    ],
    "optimizer": [
        ("--loss_fn", "v", "DiceCE", S), ("--tversky_alpha", "v", 0.5, F), ("--tversky_beta", "v", 0.5, F),
        ("--smooth_nr", "v", 1e-5, F), ("--smooth_dr", "v", 1e-5, F), ("--weight_decay", "v", 1e-5, F),
        ("--lr", "v", 4e-4, F), ("--momentum", "v", 0.9, F), ("--warmup_epochs", "v", 40, I),
    ],
    "training": [
        ("--start_epoch", "v", 0, I), ("--epochs", "v", 200, I), ("--save_ckpt_freq", "v", 20, I),
        ("--val_interval", "v", 20, I), ("--cv_fold", "v", 0, I), ("--cv_max_folds", "v", 5, I),
        ("--val_infer_overlap", "v", 0.5, F), ("--world_size", "v", 1, I), ("--local_rank", "v", -1, I),
        ("--dist_on_itp", "t"), ("--dist_url", "v", "env://", S), ("--backend", "v", "nccl", S),
        ("--resume", "v", "", S), ("--pretrained", "v", None, S),
    ],
    "misc": [
        ("--seed", "v", 13, I), ("--no_cuddn_auto_tuner", "t"), ("--anomaly_detection", "t"), ("--log_dir", "v", None, S),
        ("--no_neptune_logging", "f", "neptune_logging"), ("--save_eval_output", "t"), ("--output_dir", "v", None, S),
        ("--description", "v", None, S),
    ],
    "amd": [  # additions of this build
        ("--synthetic", "t"), ("--synthetic_steps", "v", 8, I), ("--synthetic_val_size", "l", [128], I),
        ("--compute_dtype", "v", "bf16", S), ("--flat_optimizer", "t"),
    ],
}


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(description="3-D medical segmentation on MI355X")
    for group_name, flags in _SPEC.items():
        g = parser.add_argument_group(group_name)
        for spec in flags:
            flag, kind = spec[0], spec[1]
            if kind == "v":
                g.add_argument(flag, default=spec[2], type=spec[3])
            elif kind == "l":
                g.add_argument(flag, nargs="*", default=list(spec[2]), type=spec[3])
            elif kind == "t":
                g.add_argument(flag, action="store_true", default=False)
            else:
                g.add_argument(flag, action="store_false", dest=spec[2], default=True)
    return parser


def collapse_lists(args):
    for k, v in vars(args).items():
        if isinstance(v, list):
            setattr(args, k, v[0] if len(v) == 1 else tuple(v))
    return args


def get_args(argv=None):
    return collapse_lists(build_parser().parse_args(argv))
