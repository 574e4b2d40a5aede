"""BEST-RQ self-supervised task (reference task_factory/ssl_task.py:31-180): cmvn(raw),
cmvn(auged) -> BestRQLayer -> encoder -> logits layer -> per-codebook masked / total loss."""
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
