"""BEST-RQ self-supervised task (reference task_factory/ssl_task.py:31-180): cmvn(raw),
cmvn(auged) -> BestRQLayer -> encoder -> logits layer -> per-codebook masked / total loss."""
import torch

from speech2text_amd.model.decoder.decoder import Decoder
from speech2text_amd.model.encoder.encoder import Encoder
from speech2text_amd.model.loss.loss import Loss
from speech2text_amd.model.ssl.best_rq import (BestRQLayer, BestRQLayerConfig,
                                               MaskingStrategyConfig)
from speech2text_amd.task_factory.base import TaskBase


class SslTask(TaskBase):
    def __init__(self, config) -> None:
        super().__init__(config)
        self._loss_config = config["loss"]
        assert self._loss_config["loss_select"] in ("tot_loss", "mask_loss")
        sc = config["ssl_layer"]
        self._ssl_layer = BestRQLayer(layer_config=BestRQLayerConfig(**sc["layer_config"]),
                                      masking_config=MaskingStrategyConfig(**sc["masking_config"]))
        self._encoder = Encoder(config["encoder"])
        self._logits_layer = Decoder(config["logits_layer"])
        self._loss = Loss(self._loss_config)
        from speech2text_amd.model.utils import SslMetric, SslMetricConfig
        self._metric = SslMetric(config=SslMetricConfig(**(config.get("metric") or {})))

    def training_step(self, batch, batch_idx):
        if "raw_feat" in batch:
            raw = self._global_cmvn(batch["raw_feat"])
            aug = self._global_cmvn(batch["auged_feat"])
            feat_len = batch["feat_length"]
        else:   # raw PCM on the GPU: features computed once, augmentation is upstream of us
            raw, feat_len = self.features(batch)
            aug = raw.clone()
        out = self._ssl_layer(raw, aug, feat_len)
        enc, enc_len = self._encoder(out["masked_feats"], feat_len)
        logits, logits_len = self._logits_layer(enc, enc_len)
        self.log_dict({"mask_rate": out["masked_dim"].sum() / logits_len.sum()}, sync_dist=True)
        mask_losses, tot_losses = [], []
        for cb in range(self._ssl_layer.num_codebooks):
            lab = out["labels"][cb]
            mask_losses.append(self._loss({"logits": logits, "ori_labels": lab,
                                           "mask": out["masked_dim"]}))
            tot_losses.append(self._loss({"logits": logits, "ori_labels": lab,
                                          "mask": logits_len}))
        n = self._ssl_layer.num_codebooks
        mask_loss, tot_loss = sum(mask_losses) / n, sum(tot_losses) / n
        loss = tot_loss if self._loss_config["loss_select"] == "tot_loss" else mask_loss
        self.log_dict({"train_loss": loss, "train_loss/tot_loss": tot_loss,
                       "train_loss/mask_loss": mask_loss}, sync_dist=True)
        return loss.mean()

    @torch.no_grad()
    def validation_step(self, batch, batch_idx):
        """reference ssl_task.py:182-254: the raw features stand in for the augmented ones; per
        codebook masked / total loss and top-k accuracy over the masked positions, averaged."""
        if "raw_feat" in batch:
            raw = self._global_cmvn(batch["raw_feat"])
            feat_len = batch["feat_length"]
        else:
            raw, feat_len = self.features(batch)
        out = self._ssl_layer(raw, raw, feat_len)
        enc, enc_len = self._encoder(out["masked_feats"], feat_len)
        logits, logits_len = self._logits_layer(enc, enc_len)
        n = self._ssl_layer.num_codebooks
        mask_losses, tot_losses, accs = [], [], {}
        for cb in range(n):
            lab = out["labels"][cb]
            mask_losses.append(self._loss({"logits": logits, "ori_labels": lab,
                                           "mask": out["masked_dim"]}))
            tot_losses.append(self._loss({"logits": logits, "ori_labels": lab, "mask": logits_len}))
            preds = self._loss.predict(logits)
            for k, v in self._metric(logits=preds, labels=lab, masked_dim=out["masked_dim"]).items():
                accs[k] = accs.get(k, 0.0) + v / n
        mask_loss, tot_loss = sum(mask_losses) / n, sum(tot_losses) / n
        loss = tot_loss if self._loss_config["loss_select"] == "tot_loss" else mask_loss
        info = {"val_loss": loss, "val_loss/tot_loss": tot_loss, "val_loss/mask_loss": mask_loss,
                **accs}
        self.log_dict(info, sync_dist=True, prog_bar=True, logger=True)
        return info
