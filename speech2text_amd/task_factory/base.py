"""Task plumbing shared by the *_task modules.

The reference tasks are pytorch_lightning.LightningModules driven by pl.Trainer
(build_task.py:143-148).  Lightning is not part of this stack: `TaskBase` keeps the subset of
the LightningModule surface the reference's tasks use on the training path (training_step,
configure_optimizers, log / log_dict, current_epoch / global_step) and speech2text_amd.trainer
drives it.  Data loading, tokenisers and decoding metrics are outside the accelerated path
(SURVEY.md section 8f); tasks accept batches that already follow the reference's batch-dict
contract (dataset/utils.py:182-202) or carry raw PCM for the on-GPU frontend.
"""
import copy

import torch
import torch.nn as nn

from speech2text_amd.dataset.frontend.frontend import FeatType
from speech2text_amd.model.layer.global_cmvn import GlobalCmvnLayer


class TaskBase(nn.Module):
    def __init__(self, config) -> None:
        super().__init__()
        self._dataset_config = config["dataset"]
        self._optim_config = config.get("optim_setup")
        # the reference builds the tokenizer unconditionally (ctc_task.py:49); synthetic-batch
        # harnesses (bench.py) carry token ids and no tokenizer section
        self._tokenizer = None
        if config.get("tokenizer") is not None:
            from speech2text_amd.dataset.utils import TokenizerSetup
            self._tokenizer = TokenizerSetup(config["tokenizer"])
        self._metric_config = config.get("metric")
        self._metric = None
        self._frontend = self._get_frontend(copy.deepcopy(config["dataset"]))
        self._global_cmvn = GlobalCmvnLayer(config=self._dataset_config)
        self.logged = {}
        self.current_epoch = 0
        self.global_step = 0

    @staticmethod
    def _get_frontend(config):
        if config["feat_type"] == "fbank":
            config["feat_config"]["dither"] = 0.0
        return FeatType[config["feat_type"]].value(**config["feat_config"])

    # ---- Lightning-like logging hooks (values are reduced by the trainer)
    def log(self, name, value, **kwargs):
        self.logged[name] = value

    def log_dict(self, d, **kwargs):
        self.logged.update(d)

    # ---- features: reference contract ("feat"/"feat_length") or raw PCM on the GPU
    def features(self, batch):
        if "feat" in batch:
            return self._global_cmvn(batch["feat"]), batch["feat_length"]
        pcm, n = batch["pcm"], batch["pcm_length"]
        mean = getattr(self._global_cmvn, "global_mean", None)
        istd = getattr(self._global_cmvn, "global_istd", None)
        feats, frames = self._frontend.forward_batch(pcm, n, mean, istd)
        return feats, frames

    def _asr_metric(self, predictor=None, joiner=None):
        """AsrMetric of the YAML's `metric:` section (reference ctc_task.py:56-57,
        rnnt_task.py:65-69), or None when the config carries no tokenizer / metric."""
        if self._tokenizer is None or self._metric_config is None:
            return None
        from speech2text_amd.model.utils import AsrMetric, AsrMetricConfig
        return AsrMetric(tokenizer=self._tokenizer, config=AsrMetricConfig(**self._metric_config),
                         predictor=predictor, joiner=joiner)

    def _wer(self, hidden_states, lengths, labels):
        if self._metric is None:
            raise RuntimeError("validation_step needs the YAML's `tokenizer:` and `metric:` sections "
                               "(reference task_factory/ctc_task.py:39-57)")
        return self._metric(hidden_states, lengths, labels)

    def configure_optimizers(self):
        from speech2text_amd.optimizer.optim_setup import OptimSetup
        Optimizer, LR_Scheduler = OptimSetup(self._optim_config)
        optimizer = Optimizer(self._optimizer_params(), **self._optim_config["optimizer"]["config"])
        sched = LR_Scheduler(optimizer=optimizer, **self._optim_config["lr_scheduler"]["config"])
        return {"optimizer": optimizer,
                "lr_scheduler": {"scheduler": sched,
                                 **self._optim_config["lr_scheduler"]["step_config"]}}

    def _optimizer_params(self):
        return self.parameters()
