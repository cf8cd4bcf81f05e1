// TEST INFRASTRUCTURE (oracle/cvshim): minimal g2o::SE3Quat / g2o::Sim3 so that the reference's cConverter.h / cConverter.cpp compile without g2o's
// generated config.h.  Only what cConverter touches: construction from (R, t), rotation() as a quaternion with toRotationMatrix(), translation(), scale().
#pragma once
#include <Eigen/Dense>
#include <Eigen/Geometry>
namespace g2o {
typedef Eigen::Matrix<double, 3, 1> Vector3d;
typedef Eigen::Matrix<double, 3, 3> Matrix3d;
class SE3Quat {
public:
	Eigen::Quaterniond q; Eigen::Vector3d t;
	SE3Quat() : q(Eigen::Quaterniond::Identity()), t(Eigen::Vector3d::Zero()) {}
	SE3Quat(const Eigen::Matrix3d& R, const Eigen::Vector3d& t_) : q(R), t(t_) {}
	const Eigen::Quaterniond& rotation() const { return q; }
	const Eigen::Vector3d& translation() const { return t; }
	Eigen::Matrix<double, 4, 4> to_homogeneous_matrix() const { Eigen::Matrix4d M = Eigen::Matrix4d::Identity(); M.block<3, 3>(0, 0) = q.toRotationMatrix(); M.block<3, 1>(0, 3) = t; return M; }
};
class Sim3 {
public:
	Eigen::Quaterniond q; Eigen::Vector3d t; double s;
	Sim3() : q(Eigen::Quaterniond::Identity()), t(Eigen::Vector3d::Zero()), s(1.0) {}
	Sim3(const Eigen::Matrix3d& R, const Eigen::Vector3d& t_, double s_) : q(R), t(t_), s(s_) {}
	const Eigen::Quaterniond& rotation() const { return q; }
	const Eigen::Vector3d& translation() const { return t; }
	double scale() const { return s; }
};
}  // namespace g2o
