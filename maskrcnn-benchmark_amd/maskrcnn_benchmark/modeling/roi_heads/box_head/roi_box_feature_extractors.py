"""Box-head feature extractors (reference roi_heads/box_head/roi_box_feature_extractors.py:14-151)."""
from torch import nn
from torch.nn import functional as F

from maskrcnn_benchmark.modeling import registry
from maskrcnn_benchmark.modeling.backbone import resnet
from maskrcnn_benchmark.modeling.make_layers import group_norm, make_fc
from maskrcnn_benchmark.modeling.poolers import make_pooler


@registry.ROI_BOX_FEATURE_EXTRACTORS.register("ResNet50Conv5ROIFeatureExtractor")
class ResNet50Conv5ROIFeatureExtractor(nn.Module):
    """C4 models: ROIAlign on the stride-16 map, then res5 on the pooled crops."""

    def __init__(self, config, in_channels):
        super(ResNet50Conv5ROIFeatureExtractor, self).__init__()
        self.pooler = make_pooler(config, "ROI_BOX_HEAD")
        R = config.MODEL.RESNETS
        stage = resnet.StageSpec(index=4, block_count=3, return_features=False)
        self.head = resnet.ResNetHead(block_module=R.TRANS_FUNC, stages=(stage,), num_groups=R.NUM_GROUPS,
                                      width_per_group=R.WIDTH_PER_GROUP, stride_in_1x1=R.STRIDE_IN_1X1,
                                      stride_init=None, res2_out_channels=R.RES2_OUT_CHANNELS,
                                      dilation=R.RES5_DILATION)
        self.out_channels = self.head.out_channels

    def forward(self, x, proposals):
        return self.head(self.pooler(x, proposals))


@registry.ROI_BOX_FEATURE_EXTRACTORS.register("FPN2MLPFeatureExtractor")
class FPN2MLPFeatureExtractor(nn.Module):
    """FPN models: multi-level ROIAlign (7x7), flatten, two 1024-d fully connected layers."""

    def __init__(self, cfg, in_channels):
        super(FPN2MLPFeatureExtractor, self).__init__()
        H = cfg.MODEL.ROI_BOX_HEAD
        self.pooler = make_pooler(cfg, "ROI_BOX_HEAD")
        self.fc6 = make_fc(in_channels * H.POOLER_RESOLUTION ** 2, H.MLP_HEAD_DIM, H.USE_GN)
        self.fc7 = make_fc(H.MLP_HEAD_DIM, H.MLP_HEAD_DIM, H.USE_GN)
        self.out_channels = H.MLP_HEAD_DIM

    def forward(self, x, proposals):
        x = self.pooler(x, proposals)
        x = x.flatten(1)
        return F.relu(self.fc7(F.relu(self.fc6(x))))


@registry.ROI_BOX_FEATURE_EXTRACTORS.register("FPNXconv1fcFeatureExtractor")
class FPNXconv1fcFeatureExtractor(nn.Module):
    """FPN models, conv head: N stacked 3x3 convs (+GN) and one fully connected layer."""

    def __init__(self, cfg, in_channels):
        super(FPNXconv1fcFeatureExtractor, self).__init__()
        H = cfg.MODEL.ROI_BOX_HEAD
        self.pooler = make_pooler(cfg, "ROI_BOX_HEAD")
        layers = []
        for _ in range(H.NUM_STACKED_CONVS):
            conv = nn.Conv2d(in_channels, H.CONV_HEAD_DIM, kernel_size=3, stride=1, padding=H.DILATION,
                             dilation=H.DILATION, bias=not H.USE_GN)
            nn.init.normal_(conv.weight, std=0.01)
            if not H.USE_GN:
                nn.init.constant_(conv.bias, 0)
            layers.append(conv)
            in_channels = H.CONV_HEAD_DIM
            if H.USE_GN:
                layers.append(group_norm(in_channels))
            layers.append(nn.ReLU(inplace=True))
        self.xconvs = nn.Sequential(*layers)
        self.fc6 = make_fc(H.CONV_HEAD_DIM * H.POOLER_RESOLUTION ** 2, H.MLP_HEAD_DIM, use_gn=False)
        self.out_channels = H.MLP_HEAD_DIM

    def forward(self, x, proposals):
        x = self.xconvs(self.pooler(x, proposals))
        return F.relu(self.fc6(x.flatten(1)))


def make_roi_box_feature_extractor(cfg, in_channels):
    return registry.ROI_BOX_FEATURE_EXTRACTORS[cfg.MODEL.ROI_BOX_HEAD.FEATURE_EXTRACTOR](cfg, in_channels)
