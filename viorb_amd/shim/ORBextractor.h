// viorb_amd/shim/ORBextractor.h — drop-in replacement for the reference's include/ORBextractor.h
// (ORB_SLAM2::ORBextractor, reference include/ORBextractor.h:45-111, src/ORBextractor.cc:410-470,1043-1105).
// Same class name, constructor, operator(), getters and public mvImagePyramid; the work is forwarded to
// libviorb_hip.so through the C ABI of include/viorb.h. Build the reference with this directory first on the
// include path and link libviorb_hip.so instead of compiling src/ORBextractor.cc.
//
// OpenCV is needed for the cv:: types in the signatures (the reference already depends on it). For
// compile-testing without OpenCV define VIORB_SHIM_CV_STANDIN and provide the few cv:: stand-in types of
// tests/cpp/cv_standin.h first.
#ifndef ORBEXTRACTOR_H
#define ORBEXTRACTOR_H

#include <vector>
#include <list>
#include <stdexcept>
#include <string>
#ifndef VIORB_SHIM_CV_STANDIN
#include <opencv/cv.h>
#endif
#include "viorb.h"

namespace ORB_SLAM2
{

class ORBextractor
{
public:
    enum {HARRIS_SCORE=0, FAST_SCORE=1 };

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
        : nfeatures(nfeatures), scaleFactor(scaleFactor), nlevels(nlevels), iniThFAST(iniThFAST), minThFAST(minThFAST), mHandle(0)
    {
        viorb_extractor_params p = {nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST};
        if (viorb_extractor_create(&p, 1, 0, &mHandle) != VIORB_OK)
            throw std::runtime_error(std::string("viorb_extractor_create: ") + viorb_last_error());
        mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
        mnFeaturesPerLevel.resize(nlevels);
        viorb_extractor_tables(mHandle, &mvScaleFactor[0], &mvInvScaleFactor[0], &mvLevelSigma2[0], &mvInvLevelSigma2[0], &mnFeaturesPerLevel[0]);
        viorb_extractor_max_keypoints(mHandle, &mCap);
        mvImagePyramid.bind(mHandle, nlevels);
    }
    ~ORBextractor() { viorb_extractor_destroy(mHandle); }

    // Compute the ORB features and descriptors on an image. Mask is ignored, as in the reference.
    void operator()( cv::InputArray _image, cv::InputArray _mask, std::vector<cv::KeyPoint>& _keypoints, cv::OutputArray _descriptors)
    {
        if (_image.empty()) return;                                   // reference src/ORBextractor.cc:1046-1047
        cv::Mat image = _image.getMat();
        assert(image.type() == CV_8UC1);
        int cap = mCap;                                               // exact bound for this image size (== mCap for every ordinary camera)
        viorb_extractor_max_keypoints_for(mHandle, image.cols, image.rows, &cap);
        std::vector<viorb_keypoint> k(cap);
        cv::Mat desc(cap, 32, CV_8U);
        int n = 0;
        // Every failure is surfaced: the reference has no error channel here (it cannot fail), so a GPU-side error — VIORB_ERR_CAPACITY
        // included: a truncated keypoint set must never flow into Frame unnoticed — becomes an exception with viorb_last_error().
        const int rc = viorb_extract(mHandle, image.data, image.cols, image.rows, (int)image.step, &k[0], desc.data, cap, &n);
        if (rc != VIORB_OK) throw std::runtime_error(std::string("viorb_extract: ") + viorb_last_error());
        _keypoints.clear(); _keypoints.reserve(n);
        for (int i = 0; i < n; i++)                                    // viorb_keypoint is layout-identical to cv::KeyPoint
            _keypoints.push_back(cv::KeyPoint(k[i].x, k[i].y, k[i].size, k[i].angle, k[i].response, k[i].octave, k[i].class_id));
        if (n == 0) _descriptors.release();
        else desc.rowRange(0, n).copyTo(_descriptors);
        // mvImagePyramid is public in the reference and read by Frame::ComputeStereoMatches (src/Frame.cc:653,743-760) only: the levels
        // stay on the device and a level is downloaded when (if) somebody indexes it (the mono-inertial path never does)
        mvImagePyramid.invalidate();
    }

    int inline GetLevels(){ return nlevels;}
    float inline GetScaleFactor(){ return scaleFactor;}
    std::vector<float> inline GetScaleFactors(){ return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors(){ return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares(){ return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares(){ return mvInvLevelSigma2; }

    // not in the reference: the device handle, for Frame::ComputeStereoMatches -> viorb_stereo_match
    viorb_extractor* handle() { return mHandle; }

    // Stands in for the reference's public `std::vector<cv::Mat> mvImagePyramid` (include/ORBextractor.h:86): operator[] / size() as
    // Frame::ComputeStereoMatches uses them, a level is copied from the device the first time it is indexed after an extraction.
    class LazyPyramid {
    public:
        LazyPyramid() : mHandle(0) {}
        void bind(viorb_extractor* h, int nlevels) { mHandle = h; mLevels.assign(nlevels, cv::Mat()); mFresh.assign(nlevels, 0); }
        void invalidate() { for (size_t l = 0; l < mFresh.size(); l++) mFresh[l] = 0; }
        size_t size() const { return mLevels.size(); }
        int downloads() const { return mDownloads; }
        cv::Mat& operator[](size_t l) {
            if (!mFresh[l]) {
                int w = 0, h = 0;
                if (viorb_extractor_level_download(mHandle, 0, (int)l, 0, 0, &w, &h) != VIORB_OK) throw std::runtime_error(std::string("mvImagePyramid: ") + viorb_last_error());
                mLevels[l].create(h, w, CV_8U);
                if (viorb_extractor_level_download(mHandle, 0, (int)l, 0, mLevels[l].data, &w, &h) != VIORB_OK) throw std::runtime_error(std::string("mvImagePyramid: ") + viorb_last_error());
                mFresh[l] = 1; mDownloads++;
            }
            return mLevels[l];
        }
    private:
        viorb_extractor* mHandle; std::vector<cv::Mat> mLevels; std::vector<char> mFresh; int mDownloads = 0;
    };
    LazyPyramid mvImagePyramid;                   // un-padded levels (the reference's 19-px border is never read)

protected:
    int nfeatures;
    double scaleFactor;
    int nlevels;
    int iniThFAST;
    int minThFAST;
    std::vector<int> mnFeaturesPerLevel;
    std::vector<float> mvScaleFactor;
    std::vector<float> mvInvScaleFactor;
    std::vector<float> mvLevelSigma2;
    std::vector<float> mvInvLevelSigma2;
    viorb_extractor* mHandle;
    int mCap;
private:
    ORBextractor(const ORBextractor&);            // one handle = one device context; not copyable
    ORBextractor& operator=(const ORBextractor&);
};

} //namespace ORB_SLAM

#endif
