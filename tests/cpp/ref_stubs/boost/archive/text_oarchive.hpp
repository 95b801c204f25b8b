// COMPILE-ONLY stand-in for the Boost.Serialization names the reference's headers mention (declarations only).
#pragma once
#include <cstddef>
#include <iosfwd>
#include <exception>
namespace boost {
namespace serialization {
class access {};
template <class Base, class Derived> Base& base_object(Derived&);
template <class T> struct array_wrapper { };
template <class T> array_wrapper<T> make_array(T*, std::size_t);
template <class T> array_wrapper<const T> make_array(const T*, std::size_t);
template <class Archive, class T> void split_free(Archive&, T&, unsigned int);
template <class Archive, class T> void split_member(Archive&, T&, unsigned int);
}  // namespace serialization
namespace archive {
struct archive_exception : std::exception { };
struct text_oarchive {
    explicit text_oarchive(std::ostream&, unsigned int = 0);
    template <class T> text_oarchive& operator&(const T&);
    template <class T> text_oarchive& operator<<(const T&);
    struct is_loading { enum { value = 0 }; };
    struct is_saving { enum { value = 1 }; };
};
struct text_iarchive {
    explicit text_iarchive(std::istream&, unsigned int = 0);
    template <class T> text_iarchive& operator&(T&);
    template <class T> text_iarchive& operator>>(T&);
    struct is_loading { enum { value = 1 }; };
    struct is_saving { enum { value = 0 }; };
};
}  // namespace archive
template <class... T> void ignore_unused(const T&...);
}  // namespace boost
#define BOOST_SERIALIZATION_ASSUME_ABSTRACT(T)
#define BOOST_SERIALIZATION_SPLIT_FREE(T)
#define BOOST_SERIALIZATION_SPLIT_MEMBER()
#define BOOST_CLASS_EXPORT_KEY(T)
#define BOOST_CLASS_EXPORT_GUID(T, K)
