"""Shared pieces of the config factories.  The configs are *data*: the field names are the API
the constructors read; values follow the reference's shipped configs."""
import os

from ctdd.config_dict import ConfigDict


def skeleton(save_directory, device="cuda"):
    c = ConfigDict()
    c.save_location = save_directory
    c.device = device
    c.distributed = False
    c.num_gpus = 0
    for sec in ("loss", "training", "data", "model", "optimizer", "saving", "sampler"):
        c[sec] = ConfigDict()
    c.training.update(train_step_name="Standard", clip_grad=True, warmup=0)
    c.optimizer.update(name="Adam", lr=2e-4)
    c.saving.sample_plot_path = os.path.join(save_directory, "PNGs")
    c.sampler.update(eps_ratio=1e-9, num_corrector_steps=10, corrector_step_size_multiplier=1.5,
                     corrector_entry_time=0.0, is_ordinal=True)
    return c


def image_data(c, name, S, image_size, channels, batch_size):
    c.data.update(name=name, train=True, download=True, S=S, batch_size=batch_size, shuffle=True,
                  image_size=image_size, shape=[channels, image_size, image_size], use_augm=False,
                  location="lib/datasets/")


def tau_unet(c, ch, ch_mult, channels, image_size, model_output):
    c.model.update(name="GaussianTargetRateImageX0PredEMAPaul", padding=False, ema_decay=0.9999, ch=ch,
                   num_res_blocks=2, ch_mult=ch_mult, input_channels=channels, scale_count_to_put_attn=1,
                   data_min_max=[0, 255], dropout=0.1, skip_rescale=True, time_embed_dim=ch,
                   time_scale_factor=1000, fix_logistic=False, model_output=model_output, num_heads=8,
                   attn_resolutions=[int(ch / 2)], concat_dim=image_size * image_size * channels,
                   rate_sigma=6.0, Q_sigma=512.0, time_exp=100.0, time_base=3.0)


def hollow(c, embed_dim, num_layers, mlp_dim, concat_dim, S):
    c.model.update(net_arch="bidir_transformer", nets="bidir_transformer2", use_cat=False, embed_dim=embed_dim,
                   bidir_readout="attention", use_one_hot_input=False, dropout_rate=0.1, concat_dim=concat_dim,
                   num_layers=num_layers, num_heads=8, attention_dropout_rate=0.1,
                   transformer_norm_type="prenorm", mlp_dim=mlp_dim, out_dim=None, readout_dim=S,
                   num_output_ffresiduals=2, qkv_dim=embed_dim, ema_decay=0.9999, time_scale_factor=1000,
                   log_prob="cat", fix_logistic=False)
