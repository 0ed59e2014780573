"""The multitask / end-to-end scripts' model class under its own defaults.

``from s2vt_amd.multitask import Video_Caption_Generator`` is the class the reference's
reinforce_multitask_e2e_attribute_loss.py:70-114 / multitask_e2e_attribute_s2vt.py:70-114 define: the same eighteen
constructor arguments with the same defaults (width = height = 299, channels = 3, feature_dim = 1536, label_dim = 400,
alpha = 0.2), `build_model()` / `build_loss()` returning the 6-tuples of those files, one multinomial sample per video in
build_loss (its batch is `batch_size`, :236, not the batch_size*8 of reinforcement_multisampling_tf_s2vt.py:228), and the
always-true weight-decay predicate (:222).  Everything else is model.Video_Caption_Generator; `attach_cnn(cnn)` turns the
video placeholders into the frame placeholders of those scripts.

reinforce_multitask_e2e_attribute_s2vt.py (the lambda-mixed objective, attribute head commented out, :58-60) is the same
class built with label_dim=0: its build_model / build_loss return the 5- / 4-tuples (:226, :375).
"""
from __future__ import annotations

from . import model as _model


class Video_Caption_Generator(_model.Video_Caption_Generator):
    def __init__(self, dim_image, n_words, word_dim, lstm_dim, batch_size, n_lstm_steps, n_video_lstm_step,
                 n_caption_lstm_step, bias_init_vector=None, loss_weight=1, decay_value=0.00005, dropout_rate=0.9,
                 width=299, height=299, channels=3, feature_dim=1536, label_dim=400, alpha=0.2, device="cuda", seed=1234,
                 multisample=1):
        super().__init__(dim_image, n_words, word_dim, lstm_dim, batch_size, n_lstm_steps, n_video_lstm_step,
                         n_caption_lstm_step, bias_init_vector=bias_init_vector, loss_weight=loss_weight, decay_value=decay_value,
                         dropout_rate=dropout_rate, width=width, height=height, channels=channels,
                         feature_dim=feature_dim if label_dim else None, label_dim=label_dim, alpha=alpha, device=device, seed=seed,
                         multisample=multisample)
        self.decay_all_variables = True          # `if 'bias' or 'BatchNorm' not in v.name` (:222): every trainable variable
